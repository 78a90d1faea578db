"""The kernels promise IEEE float32 arithmetic in source order (no contraction, correctly rounded division and
square root).  tests/gpu_probe/probe evaluates the building blocks of srt_core.h on the GPU and on the host from
the same source and compares them bit for bit on a million random inputs each."""
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
