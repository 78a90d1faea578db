"""The kernels promise IEEE float32 arithmetic in source order (no contraction, correctly rounded division and
square root).  tests/gpu_probe/probe evaluates the building blocks of srt_core.h on the GPU and on the host from
the same source and compares them bit for bit on a million random inputs each; tests/gpu_probe/exhaustive runs the
device's short forms of reciprocal, 1 / sqrt and the Smith term on all 2^32 float bit patterns."""
import os
import re
import subprocess

import pytest

import helpers

pytestmark = pytest.mark.gpu


def test_device_arithmetic_matches_host_bit_for_bit():
    exe = os.path.join(helpers.ROOT, "tests", "gpu_probe", "probe")
    assert os.path.exists(exe), "build it with `make -C tests/gpu_probe` (done by __graft_entry__.build())"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"contract_violations=(\d+)", out.stdout)
    assert m and int(m.group(1)) == 0, out.stdout
    # the double-precision helpers (atan2 of the fresnel term, x^5 of the GGX F term) may differ from libm in the
    # last bit of a double; after rounding to float they must agree in all but a vanishing fraction of inputs
    for name in ("fresnel", "point_light", "pow5", "atan2"):
        mm = re.search(r"^%s mismatches=(\d+) of (\d+)" % name, out.stdout, flags=re.M)
        assert mm and int(mm.group(1)) <= int(mm.group(2)) // 100000, out.stdout


def test_short_forms_are_ieee_on_every_float():
    """rcp3_rn, rcp_sqrt_rn and smith_term (srt_core.h) against the compiler's general IEEE sequences on the device, bit
    for bit on all 2^32 inputs, and against the host's IEEE operations through a checksum over the same inputs."""
    exe = os.path.join(helpers.ROOT, "tests", "gpu_probe", "exhaustive")
    assert os.path.exists(exe), "build it with `make -C tests/gpu_probe` (done by __graft_entry__.build())"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert re.search(r"^exhaustive_failures=0$", out.stdout, flags=re.M), out.stdout
    for name in ("rcp3_rn", "rcp_sqrt_rn", "smith_term"):
        m = re.search(r"^%s mismatches=(\d+) checksum_device=([0-9a-f]+) checksum_host=([0-9a-f]+)" % name, out.stdout, flags=re.M)
        assert m and int(m.group(1)) == 0 and m.group(2) == m.group(3), out.stdout
