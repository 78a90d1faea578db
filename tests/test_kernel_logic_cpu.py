"""The kernel source (srt_core.h + srt_scene.cpp) executed lane by lane on the CPU (tests/emu) against the oracle.

This does not replace the GPU parity tests (test_gpu_parity.py): it checks, in a container without a GPU, that
the state machine, both traversals and both BVH builders reproduce the oracle's images and ray counts.
"""
import pytest

import parity_cases
import simple_raytracer_amd as srt


@pytest.fixture(scope="module")
def emu(emu_lib):
    return srt.Renderer(library=emu_lib, strict=False)


@pytest.mark.parametrize("case", parity_cases.CASES, ids=[c[0] for c in parity_cases.CASES])
def test_kernel_source_matches_oracle(emu, oracle, case):
    lsb, npx = parity_cases.assert_case(emu, oracle, case)
    assert (lsb, npx) == (0, 0)   # same compiler family and libm on both sides: not even 1 LSB


def test_tile_layout(emu, oracle):
    assert parity_cases.tiles_roundtrip(emu, oracle) == (0, 0)


def test_rects_are_rejected_for_bvhnode_worlds(emu):
    import helpers
    world, lights, cam = helpers.default_spec(with_rect=True).host(64, 36)
    with pytest.raises(srt.SrtError) as e:
        emu.set_scene(world, lights, cam, world_kind=srt.WORLD_BVHNODE)
    assert e.value.code == srt._native.E_REFERENCE


def test_empty_world_renders_sky(emu, oracle):
    import helpers
    spec = helpers.SceneSpec()
    spec.materials = [((0.5, 0.5, 0.5), 0, 0.5, (0, 0, 0))]
    world, lights, cam = spec.host(32, 18)
    emu.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
    img = emu.render(32, 18, samples=1, max_depth=4)
    ref, _ = spec.oracle_scene().render(spec.oracle_camera(32, 18), 32, 18, samples=1, max_depth=4,
                                        world_kind=oracle.WORLD_SURFACELIST)
    assert (img == ref).all()


import mesh_cases  # noqa: E402


@pytest.mark.parametrize("case", mesh_cases.CASES, ids=[c[0] for c in mesh_cases.CASES])
def test_sah_tree_matches_linear_scan(emu, oracle, case):
    lsb, npx, st, ost = mesh_cases.run(emu, oracle, case)
    assert (lsb, npx) == (0, 0)
    for k in ("rays_path", "rays_shadow", "hit_nodes"):
        assert st[k] == ost[k], (case[0], k)
    assert st["bvh_nodes"] > 100 and st["box_tests"] > 0


import random_scenes  # noqa: E402


@pytest.mark.parametrize("seed", range(24))
def test_random_scene_matches_oracle(emu, oracle, seed):
    lsb, npx, same_counts, what = random_scenes.run(emu, oracle, seed)
    assert (lsb, npx) == (0, 0) and same_counts, what
