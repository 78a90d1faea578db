"""The kernel source (srt_core.h + srt_scene.cpp) executed lane by lane on the CPU (tests/emu) against the oracle.

This does not replace the GPU parity tests (test_gpu_parity.py): it checks, in a container without a GPU, that
the state machine, both traversals and both BVH builders reproduce the oracle's images and ray counts.
"""
import pytest

import parity_cases
import simple_raytracer_amd as srt
from simple_raytracer_amd.distributed import tile_position


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


def test_surface_order_must_name_every_surface_once(emu):
    """srt_set_surface_order (world.surfaces as one ordered list, RaytracingClasses.cs:340): the commit refuses an order that
    leaves a surface out, names one twice or names one that does not exist; n = 0 goes back to triangles, spheres, rects."""
    import ctypes as C
    import numpy as np
    import helpers
    world, lights, cam = helpers.default_spec().host(32, 18)     # five spheres
    emu.set_scene(world, lights, cam, world_kind=srt.WORLD_BVHNODE)
    lib, ctx = emu._lib, emu._ctx

    def commit_with(types, indices):
        t, i = np.array(types, dtype=np.uint8), np.array(indices, dtype=np.uint32)
        assert lib.srt_set_surface_order(ctx, t.ctypes.data_as(C.POINTER(C.c_uint8)), i.ctypes.data_as(C.POINTER(C.c_uint32)), len(t)) == 0
        return lib.srt_commit_scene(ctx, srt.WORLD_BVHNODE)

    assert commit_with([1] * 5, [4, 3, 2, 1, 0]) == 0
    assert commit_with([1] * 4, [0, 1, 2, 3]) == srt._native.E_INVALID          # one surface left out
    assert commit_with([1] * 5, [0, 1, 2, 3, 3]) == srt._native.E_INVALID       # one named twice
    assert commit_with([1] * 5, [0, 1, 2, 3, 5]) == srt._native.E_INVALID       # no such sphere
    assert commit_with([1, 1, 1, 1, 0], [0, 1, 2, 3, 0]) == srt._native.E_INVALID   # no such triangle
    assert commit_with([1, 1, 1, 1, 3], [0, 1, 2, 3, 4]) == srt._native.E_INVALID   # no such type
    assert lib.srt_set_surface_order(ctx, None, None, 0) == 0
    assert lib.srt_commit_scene(ctx, srt.WORLD_BVHNODE) == 0


def test_surface_order_changes_the_reference_tree(emu, oracle):
    """The same surfaces added in another order give another BVHNode tree (the constructor splits world.surfaces by
    position, RaytracingClasses.cs:387-425): the order handed over through srt_set_surface_order must reach the builder."""
    import ctypes as C
    import numpy as np
    import helpers
    orders = {}
    for kind in (None, "spheres-first", "interleaved"):
        spec = helpers.mesh_spec(n=8) if kind is None else helpers.reordered_spec(kind)
        world, lights, cam = spec.host(32, 18)
        emu.set_scene(world, lights, cam, world_kind=srt.WORLD_BVHNODE)
        emu.render(32, 18, samples=1, max_depth=1, counted=True)
        st = emu.stats()
        osc = spec.oracle_scene()
        assert (st["bvh_nodes"], st["bvh_depth"]) == (osc.bvh_nodes(), osc.bvh_depth())
        orders[kind] = (st["bvh_nodes"], st["bvh_depth"], st["box_tests"])
    assert len(set(orders.values())) == 3, orders     # three orders, three different trees


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


@pytest.mark.parametrize("width,height,samples,scatter,rank,world", [
    (64, 36, 1, 0, 0, 1), (64, 36, 1, 2, 0, 1), (64, 36, 5, 2, 0, 1), (70, 45, 4, 1, 0, 1), (70, 45, 3, 3, 0, 1),
    (200, 120, 3, 2, 1, 3), (33, 17, 2, 2, 0, 2), (640, 360, 3, 2, 0, 1)])
def test_item_order_covers_every_pixel_and_sample_once(emu, width, height, samples, scatter, rank, world):
    """Whatever the order of the work items (blocks first, the last samples scattered), a pass visits every
    (pixel, sample) of the rank once, and every slot of its sample buffer at most once."""
    import ctypes as C

    import helpers
    import numpy as np
    w, l, cam = helpers.default_spec().host(width, height)
    emu.set_scene(w, l, cam)
    kw = dict(samples=samples, max_depth=2)
    if world > 1:
        kw.update(layout=1, tile_w=32, tile_h=32, rank=rank, world_size=world)
    p = emu.params(width, height, **kw)
    fn = emu._lib.srt_emu_item_order
    fn.restype = C.c_int64
    cap = 1 << 22
    slots, pixels = np.empty(cap, np.uint32), np.empty(cap, np.uint32)
    n = fn(emu._ctx, C.byref(p), C.c_uint32(scatter), slots.ctypes.data_as(C.c_void_p), pixels.ctypes.data_as(C.c_void_p), C.c_uint64(cap))
    assert n > 0 and n % samples == 0
    slots, pixels = slots[:n], pixels[:n]
    real = slots != 0xFFFFFFFF
    n_work = n // samples
    assert len(np.unique(slots[real])) == real.sum()                      # no slot twice
    assert slots[real].max() < n
    sample_of = slots[real] // n_work
    pairs = pixels[real].astype(np.uint64) * samples + sample_of
    assert len(np.unique(pairs)) == len(pairs)                            # no (pixel, sample) twice
    if world == 1:
        assert len(pairs) == width * height * samples                     # and none missing
    else:
        tiles_x, tiles_y = (width + 31) // 32, (height + 31) // 32
        mine = 0
        for t in range(rank, tiles_x * tiles_y, world):
            ty, tx = tile_position(t, tiles_x)
            mine += (min(32, width - tx * 32)) * (min(32, height - ty * 32))
        assert len(pairs) == mine * samples


def test_threaded_sah_build_lays_out_the_same_tree(emu, monkeypatch):
    """The SAH builder hands large left halves to other threads and splices the pieces back: the flattened scene
    (node order, primitive order, boxes) must be byte for byte what one thread builds."""
    import ctypes as C
    import helpers
    digest = emu._lib.srt_emu_scene_digest
    digest.restype = C.c_uint64
    world, lights, cam = helpers.mesh_spec(n=40).host(64, 36)       # 6,400 + 72 triangles, 3 spheres
    seen = {}
    for fork_min in ("0", "101", "500", "3000"):
        monkeypatch.setenv("SRT_BUILD_FORK_MIN", fork_min)
        emu.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
        st = emu.stats()
        seen[fork_min] = (digest(emu._ctx), st["bvh_nodes"], st["bvh_depth"], st["n_prims"])
    assert seen["0"][0] != 0
    assert len(set(seen.values())) == 1, seen


def test_thread_built_tree_of_a_large_mesh_matches_linear_scan(emu, oracle):
    import mesh_cases
    lsb, npx, colours, nodes = mesh_cases.large_mesh_window(emu, oracle)
    assert (lsb, npx) == (0, 0)
    assert colours > 20 and nodes > 10000


@pytest.mark.parametrize("n_devices", [1, 2, 3, 8])
@pytest.mark.parametrize("size,tile", [((200, 120), 0), ((70, 45), 16), ((33, 17), 32)])
def test_frame_split_over_virtual_devices_is_the_single_device_frame(emu, emu_lib, n_devices, size, tile):
    """The host orchestration of a context with several devices (plan, per-device tiles, gather, scatter into the
    frame: srt_multi.h, the code libsrt.so runs over its GPUs) on the harness's virtual devices: the frame and the ray
    counts do not depend on the number of devices, including devices that own no tile (33 x 17 is one 32 x 32 tile...
    two with the edge; 8 devices leave six without work)."""
    import helpers
    width, height = size
    world, lights, cam = helpers.default_spec().host(width, height)
    emu.set_scene(world, lights, cam)
    ref = emu.render(width, height, samples=3, max_depth=4, seed=7)
    ref_st = emu.stats()
    multi = srt.Renderer(library=emu_lib, strict=False, devices=[0] * n_devices)
    assert len(multi.device_info()["devices"]) == n_devices
    multi.set_scene(world, lights, cam)
    img = multi.render(width, height, samples=3, max_depth=4, seed=7, tile_w=tile, tile_h=tile)
    st = multi.stats()
    multi.close()
    assert (img == ref).all()
    for k in ("rays_path", "rays_shadow", "hit_nodes", "box_tests", "sphere_tests"):
        assert st[k] == ref_st[k], k


def test_a_context_with_several_devices_partitions_the_frame_itself(emu_lib):
    import helpers
    multi = srt.Renderer(library=emu_lib, strict=False, devices=[0, 0])
    world, lights, cam = helpers.default_spec().host(64, 36)
    multi.set_scene(world, lights, cam)
    with pytest.raises(srt.SrtError) as e:
        multi.render(64, 36, samples=1, layout=srt.LAYOUT_TILES, tile_w=32, tile_h=32, rank=1, world_size=2)
    assert e.value.code == srt._native.E_INVALID
    multi.close()


def test_a_rank_of_several_cannot_render_a_row_major_frame(emu):
    """Only the rank's own tiles would be written: the rest of the frame would be whatever the buffer held."""
    import ctypes as C
    p = emu.params(64, 36, layout=srt.LAYOUT_FRAME, rank=0, world_size=2)
    assert emu._lib.srt_local_pixels(C.byref(p)) == srt._native.E_INVALID


import reference_tree  # noqa: E402


@pytest.mark.parametrize("name", sorted(reference_tree.SCENES))
def test_reference_tree_is_the_oracles_tree(emu, oracle, name):
    """Shape (nodes, depth, leaf order) and test counts of the BVHNode world against the oracle's pointer tree."""
    import ctypes as C
    import numpy as np
    st, ost, mine, theirs, lsb, osc = reference_tree.run(emu, oracle, name)
    assert lsb == 0
    assert mine == theirs, (mine, theirs)
    unasked = reference_tree.check_counts(st, ost)
    assert (unasked > 0) == (name in reference_tree.STACK_WALKED), (name, unasked)   # the scenes that leave shadow rays out did so
    fn = emu._lib.srt_emu_leaf_order
    fn.restype = C.c_int64
    order = np.empty(st["n_prims"], np.uint32)
    n = fn(emu._ctx, order.ctypes.data_as(C.c_void_p), C.c_uint64(len(order)))
    assert n == st["n_prims"]
    assert list(reference_tree.insertion_ranks(name, order)) == list(reference_tree.insertion_ranks(name, osc.bvh_leaf_order(), oracle_side=True))


INERT_KINDS = ("plain", "mirror", "rough", "huge-brightness", "huge-light", "nan-metal")


@pytest.mark.parametrize("world_kind", ["bvhnode", "surfacelist"])
@pytest.mark.parametrize("kind", INERT_KINDS)
def test_shadow_rays_that_cannot_matter_are_the_oracles(emu, oracle, kind, world_kind):
    """A light behind the surface adds exactly nothing whatever its shadow ray meets (RaytracingClasses.cs:113-185 with
    NdotL = 0), so stack-walked scenes leave that ray out.  The picture must stay the oracle's bit for bit, the rays left
    out must be exactly the ones the oracle's restatement of the rule marks, and scenes or materials outside the rule's
    conditions (no roughness, values that are not of ordinary size) must keep their shadow rays."""
    import helpers
    spec = helpers.inert_light_spec(kind)
    w, h = 56, 32
    world, lights, cam = spec.host(w, h)
    wk = srt.WORLD_BVHNODE if world_kind == "bvhnode" else srt.WORLD_SURFACELIST
    emu.set_scene(world, lights, cam, world_kind=wk)
    img = emu.render(w, h, samples=2, max_depth=4, seed=9, counted=True)
    st = emu.stats()
    ref, ost = spec.oracle_scene().render(spec.oracle_camera(w, h), w, h, samples=2, max_depth=4, seed=9,
                                          world_kind=oracle.WORLD_BVHNODE if world_kind == "bvhnode" else oracle.WORLD_SURFACELIST)
    lsb, _ = helpers.max_lsb_diff(img, ref)
    assert lsb == 0
    unasked = st["shadow_unasked"]
    assert unasked == ost["shadow_inert"], (unasked, ost["shadow_inert"])
    assert st["rays_shadow"] == ost["rays_shadow"]
    if kind in ("plain", "rough"):
        assert 0 < unasked < st["rays_shadow"]
    else:
        assert unasked == 0
