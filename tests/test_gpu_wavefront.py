"""The level-by-level tracer (csrc/srt_wavefront.hip) against the oracle and against the one-kernel tracer.

Scenes whose tree is walked with a stack switch to it once a render has shown more than SRT_WAVEFRONT_MIN_RAYS rays per
sample; the tests set that threshold to 0 to run it from the first render, and SRT_WAVEFRONT_POOL_MB to make its node pool
small.  Both knobs are read per render / per context, so one process can compare the tracers.  Everything it computes
goes through the same per-node functions as the other kernels: pictures must be equal bit for bit and the ray counts
must be the oracle's."""
import numpy as np
import pytest

import helpers
import mesh_cases
import simple_raytracer_amd as srt

pytestmark = pytest.mark.gpu


@pytest.fixture()
def forced(monkeypatch):
    monkeypatch.setenv("SRT_WAVEFRONT_MIN_RAYS", "0")


def deep_tree_spec(n=8, spheres=60):
    """A mesh inside a cloud of half-metallic spheres: diffuse, metal and reflect child at every hit (up to 3,280 nodes per
    sample at depth 8)."""
    spec = helpers.mesh_spec(n=n)
    rng = np.random.default_rng(3)
    for _ in range(spheres):
        c = rng.uniform(-3, 3, 3) + np.array([0, 0, -6.0])
        spec.materials.append((tuple(float(v) for v in rng.uniform(0.3, 1, 3)), 0.5, 0.9, (0, 0, 0)))
        spec.spheres.append((tuple(float(v) for v in c), float(rng.uniform(0.2, 0.6)), len(spec.materials) - 1))
    return spec


@pytest.mark.parametrize("case", mesh_cases.CASES, ids=[c[0] for c in mesh_cases.CASES])
def test_mesh_scenes_match_the_oracle(gpu_renderer_factory, oracle, forced, case):
    r = gpu_renderer_factory()
    lsb, npx, st, ost = mesh_cases.run(r, oracle, case, counted=True)
    r.close()
    _, _, w, h, spp, _, _ = case
    assert (lsb, npx) == (0, 0)
    for k in ("rays_path", "rays_shadow", "hit_nodes"):
        assert st[k] == ost[k], k
    assert st["wavefront_items"] >= w * h * spp and st["wavefront_given_up"] == 0      # every sample went through it (the count includes padding pixels)
    assert st["wavefront_nodes"] >= st["rays_path"]                                      # one record per GetRayColor call, plus holes
    assert st["box_tests"] > 0 and st["tri_tests"] > 0


def test_deep_trees_match_the_oracle_and_the_other_tracer(gpu_renderer_factory, oracle, monkeypatch):
    spec = deep_tree_spec()
    w, h = 96, 54
    world, lights, cam = spec.host(w, h)
    ref, ost = spec.oracle_scene().render(spec.oracle_camera(w, h), w, h, samples=2, max_depth=8, seed=9, world_kind=oracle.WORLD_SURFACELIST)
    r = gpu_renderer_factory()
    r.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
    # default threshold: nothing is known about a new scene, so a blocking render traces the first sample of every pixel
    # with the one-kernel tracer, reads the ray count, and gives the other samples to the tracer that count calls for
    first = r.render(w, h, samples=2, max_depth=8, seed=9, counted=True)
    st1 = r.stats()
    assert st1["wavefront_items"] >= w * h and st1["rays_path"] > 5 * 2 * w * h
    second = r.render(w, h, samples=2, max_depth=8, seed=9, counted=True)
    st2 = r.stats()
    assert st2["wavefront_items"] == 2 * st1["wavefront_items"] and st2["wavefront_given_up"] == 0
    assert st2["wavefront_twins"] > 0            # reported, but a counting render walks the tree for them all the same: the test counts below
    for img, st in ((first, st1), (second, st2)):
        assert helpers.max_lsb_diff(img, ref) == (0, 0)
        for k in ("rays_path", "rays_shadow", "hit_nodes"):
            assert st[k] == ost[k], k
    # the same traversals, so the same tests (a box test counts once per child box in either tracer)
    for k in ("box_tests", "sphere_tests", "tri_tests", "rect_tests", "shadow_unasked"):
        assert st1[k] == st2[k], k
    # and back: the knob that turns it off
    monkeypatch.setenv("SRT_NO_WAVEFRONT", "1")
    third = r.render(w, h, samples=2, max_depth=8, seed=9)
    assert r.stats()["wavefront_items"] == 0 and np.array_equal(third, ref)
    r.close()


def test_a_pool_that_is_too_small_hands_the_chunk_to_the_other_tracer(gpu_renderer_factory, oracle, forced, monkeypatch):
    """32 MiB hold about 160,000 node records.  Before anything is known about a scene a chunk is sized for 24 records per
    sample; these trees need far more, so chunks are given up and traced by the stand-by kernel: the picture and the counts
    stay the oracle's.  The next render sizes its chunks by what was seen."""
    monkeypatch.setenv("SRT_WAVEFRONT_POOL_MB", "32")
    spec = deep_tree_spec(spheres=140)
    for i in range(len(spec.materials)):     # every surface fully smooth: no branch of a ray tree ends early
        a, m, s, e = spec.materials[i]
        spec.materials[i] = (a, 0.5, 1.0, e)
    w, h = 128, 72
    world, lights, cam = spec.host(w, h)
    ref, ost = spec.oracle_scene().render(spec.oracle_camera(w, h), w, h, samples=1, max_depth=8, seed=4, world_kind=oracle.WORLD_SURFACELIST)
    r = gpu_renderer_factory()
    r.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
    img = r.render(w, h, samples=1, max_depth=8, seed=4, counted=True)
    st = r.stats()
    assert st["wavefront_given_up"] > 0, st
    assert helpers.max_lsb_diff(img, ref) == (0, 0)
    for k in ("rays_path", "rays_shadow", "hit_nodes"):
        assert st[k] == ost[k], k
    again = r.render(w, h, samples=1, max_depth=8, seed=4, counted=True)
    st2 = r.stats()
    assert np.array_equal(again, img)
    for k in ("rays_path", "rays_shadow", "hit_nodes"):
        assert st2[k] == ost[k], k
    r.close()


def test_given_up_chunks_never_read_what_they_could_not_write(gpu_renderer_factory, oracle, forced, monkeypatch):
    """A camera that looks nowhere sends NaN rays: every node has three children and every surface is tested, so 3^6 nodes
    per sample outgrow a 32 MiB pool at once.  Entries that an overflowing kernel could not write hold whatever the pool
    held before (here: the records of another scene's render), so the later kernels of such a chunk must not touch the
    arrays at all; the stand-by kernel traces the chunk.  Found by tools/degenerate_fuzz.py."""
    monkeypatch.setenv("SRT_WAVEFRONT_POOL_MB", "32")
    r = gpu_renderer_factory()
    # fill the pool with the records of an ordinary render
    spec = deep_tree_spec()
    r.set_scene(*spec.host(160, 90), world_kind=srt.WORLD_SURFACELIST)
    r.render(160, 90, samples=2, max_depth=8, seed=2)
    assert r.stats()["wavefront_items"] > 0
    nowhere = helpers.mesh_spec(n=6)
    nowhere.materials = [(a, 0.5, 1.0, e) for a, m, sm, e in nowhere.materials]     # no branch of a ray tree ends early
    o, _, fov = nowhere.camera
    nowhere.camera = (o, o, fov)                 # lookAt == position: the camera basis is NaN
    w, h = 40, 24
    r.set_scene(*nowhere.host(w, h), world_kind=srt.WORLD_SURFACELIST)
    img = r.render(w, h, samples=3, max_depth=6, seed=5, counted=True)
    st = r.stats()
    ref, ost = nowhere.oracle_scene().render(nowhere.oracle_camera(w, h), w, h, samples=3, max_depth=6, seed=5, world_kind=oracle.WORLD_SURFACELIST)
    r.close()
    assert st["wavefront_given_up"] > 0
    assert helpers.max_lsb_diff(img, ref)[0] <= 1
    for k in ("rays_path", "rays_shadow", "hit_nodes"):
        assert st[k] == ost[k], k


def test_one_deep_tree_is_enough(gpu_renderer_factory, oracle):
    """Two small glossy spheres beside a mesh: few rays per sample on average, but every sample that meets a sphere grows a
    tree of hundreds of calls, which one lane of the one-kernel tracer would walk node after node.  The probe sample reports
    the largest tree it saw, and that alone sends the scene to the level-by-level tracer."""
    spec = helpers.mesh_spec(n=8, with_spheres=False)
    spec.materials.append(((0.9, 0.8, 0.7), 0.5, 1.0, (0, 0, 0)))
    spec.spheres = [((-2.2, 0.4, -5.0), 0.35, len(spec.materials) - 1), ((2.4, -0.6, -5.5), 0.3, len(spec.materials) - 1)]
    w, h = 128, 72
    ref, ost = spec.oracle_scene().render(spec.oracle_camera(w, h), w, h, samples=3, max_depth=8, seed=3, world_kind=oracle.WORLD_SURFACELIST)
    assert ost["rays_path"] < 5 * 3 * w * h                      # the average would not have done it
    r = gpu_renderer_factory()
    r.set_scene(*spec.host(w, h), world_kind=srt.WORLD_SURFACELIST)
    img = r.render(w, h, samples=3, max_depth=8, seed=3)
    st = r.stats()
    r.close()
    assert helpers.max_lsb_diff(img, ref) == (0, 0) and st["rays_path"] == ost["rays_path"]
    assert st["wavefront_items"] >= 2 * w * h
    # the spheres are half metal: their metalness and reflectivity passes follow one ray, which is traced once
    assert 0 < st["wavefront_twins"] < st["rays_path"] // 3


def test_asynchronous_first_render_switches_after_its_probe_sample(gpu_renderer_factory, oracle):
    """srt_render_async of a new scene queues one sample per pixel; srt_poll (or srt_wait) finds it done, reads the ray count
    and queues the other samples for the tracer the count calls for.  The frame is the blocking render's."""
    import time
    spec = deep_tree_spec()
    w, h = 96, 54
    ref, ost = spec.oracle_scene().render(spec.oracle_camera(w, h), w, h, samples=3, max_depth=8, seed=9, world_kind=oracle.WORLD_SURFACELIST)
    for use_poll in (True, False):
        r = gpu_renderer_factory()
        r.set_scene(*spec.host(w, h), world_kind=srt.WORLD_SURFACELIST)
        p = r.params(w, h, samples=3, max_depth=8, seed=9)
        out = np.zeros((h, w), dtype=np.uint32)
        r.render_async(p, out)
        if use_poll:
            deadline = time.time() + 60
            while not r.poll():
                assert time.time() < deadline
                time.sleep(0.001)
        else:
            r.wait()
        st = r.stats()
        r.close()
        assert helpers.max_lsb_diff(out, ref) == (0, 0)
        assert st["wavefront_items"] >= 2 * w * h and st["rays_path"] == ost["rays_path"]       # samples 1 and 2 of every pixel


def test_contexts_with_several_devices_use_it_on_every_device(oracle):
    """A context over the one GPU named twice: the first frame measures the scene on both replicas, the second is traced
    level by level on both (each replica has its own pool); same picture as the oracle's."""
    spec = deep_tree_spec()
    w, h = 128, 72
    ref, ost = spec.oracle_scene().render(spec.oracle_camera(w, h), w, h, samples=2, max_depth=8, seed=9, world_kind=oracle.WORLD_SURFACELIST)
    r = srt.Renderer(devices=[0, 0], gather=srt.GATHER_PEER_COPY)
    r.set_scene(*spec.host(w, h), world_kind=srt.WORLD_SURFACELIST)
    first = r.render(w, h, samples=2, max_depth=8, seed=9)
    assert r.stats()["wavefront_items"] == 0
    second = r.render(w, h, samples=2, max_depth=8, seed=9)
    st = r.stats()
    per_device = [r.device_stats(i)["wavefront_items"] for i in range(2)]
    r.close()
    assert np.array_equal(first, ref) and np.array_equal(second, ref)
    assert all(n > 0 for n in per_device) and st["wavefront_items"] == sum(per_device) and st["rays_path"] == ost["rays_path"]


def test_four_lights_and_shallow_trees_size_their_chunks_by_the_shadow_queue(gpu_renderer_factory, oracle, forced, monkeypatch):
    """With four lights all around and depth 2 a level has more shadow queries than the chunk has node records; the queue
    shares the pool's capacity, so the chunks are sized by it once it has been seen: a render may give chunks up (they are
    traced by the stand-by kernel), but renders after a clean one stay clean instead of alternating.  32 MiB hold about
    160,000 entries for 294,912 samples; the camera looks down at the floor, so every sample asks about every light."""
    monkeypatch.setenv("SRT_WAVEFRONT_POOL_MB", "32")
    spec = helpers.mesh_spec(n=8)
    spec.camera = ((0, 0, 0), (0.0, 1.9, -5.0), 30)
    spec.lights = [((4.0, -3.0, 1.0), 5.0, None), ((-3.0, -2.0, -2.0), 2.0, (1.2, 0.4, 0.4)),
                   ((0.0, -4.0, 2.0), 3.0, None), ((-4.0, -3.0, 3.0), 2.0, (0.4, 0.4, 1.2))]
    w, h = 512, 288
    ref, ost = spec.oracle_scene().render(spec.oracle_camera(w, h), w, h, samples=2, max_depth=2, seed=6, world_kind=oracle.WORLD_SURFACELIST)
    assert ost["rays_shadow"] > 2 * ost["rays_path"]          # the case: more queries than nodes
    r = gpu_renderer_factory()
    r.set_scene(*spec.host(w, h), world_kind=srt.WORLD_SURFACELIST)
    given_up = []
    for _ in range(5):
        img = r.render(w, h, samples=2, max_depth=2, seed=6)
        st = r.stats()
        assert helpers.max_lsb_diff(img, ref) == (0, 0) and st["rays_path"] == ost["rays_path"] and st["rays_shadow"] == ost["rays_shadow"]
        given_up.append(st["wavefront_given_up"])
    r.close()
    clean = given_up.index(0)                                  # some render is clean ...
    assert all(g == 0 for g in given_up[clean:]), given_up     # ... and so is every render after it


def test_tiles_passes_and_padding_pixels(gpu_renderer_factory, forced, monkeypatch):
    """Tile layout with edge tiles (padding pixels are dead nodes), several sample passes (a small sample buffer) and a rank
    that owns a part of the tiles: the same pixels as the one-kernel tracer."""
    monkeypatch.setenv("SRT_SAMPLE_BUFFER_MB", "1")
    W, H = 400, 200          # 13 x 13 tiles, 56 of them (28,672 pixels, 344 KB per sample) for rank 1: two passes of 3 + 2 samples
    world, lights, cam = mesh_cases.teapot_with_spheres(W, H)
    kw = dict(samples=5, max_depth=5, seed=11, layout=srt.LAYOUT_TILES, tile_w=32, tile_h=16, rank=1, world_size=3)
    r = gpu_renderer_factory()
    r.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
    a = r.render(W, H, **kw)
    sa = r.stats()
    r.close()
    monkeypatch.setenv("SRT_NO_WAVEFRONT", "1")
    r = gpu_renderer_factory()
    r.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
    b = r.render(W, H, **kw)
    sb = r.stats()
    r.close()
    assert sa["wavefront_items"] > 0 and sb["wavefront_items"] == 0
    assert np.array_equal(a, b)
    for k in ("rays_path", "rays_shadow", "hit_nodes"):
        assert sa[k] == sb[k], k


def test_scenes_it_does_not_serve_keep_their_tracer(gpu_renderer_factory, forced):
    """Reference trees (SRT_WORLD_BVHNODE), debug modes and scenes without lights are not its business."""
    world, lights, cam = mesh_cases.sphere_mesh(64, 36)
    r = gpu_renderer_factory()
    r.set_scene(world, lights, cam, world_kind=srt.WORLD_BVHNODE)
    r.render(64, 36, samples=1, max_depth=4, seed=1)
    assert r.stats()["wavefront_items"] == 0
    r.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
    r.render(64, 36, samples=1, max_depth=4, seed=1, mode=srt.MODE_NORMAL)
    assert r.stats()["wavefront_items"] == 0
    r.render(64, 36, samples=1, max_depth=0, seed=1)
    assert r.stats()["wavefront_items"] == 0
    r.render(64, 36, samples=1, max_depth=4, seed=1)
    assert r.stats()["wavefront_items"] >= 64 * 36
    r.close()


def test_level_by_level_render_can_be_recorded_into_a_graph(oracle):
    """srt_render_device of a scene that the level-by-level tracer serves, recorded by stream capture once the pool exists
    and the scene's ray count is known (two renders outside the capture): the chunk's memsets and its few dozen kernels
    become the graph, every replay writes the blocking render's picture, and the context works as before afterwards."""
    import torch
    spec = deep_tree_spec()
    w, h = 96, 54
    ref, ost = spec.oracle_scene().render(spec.oracle_camera(w, h), w, h, samples=2, max_depth=8, seed=9, world_kind=oracle.WORLD_SURFACELIST)
    r = srt.Renderer(device=0)
    r.set_scene(*spec.host(w, h), world_kind=srt.WORLD_SURFACELIST)
    assert np.array_equal(r.render(w, h, samples=2, max_depth=8, seed=9), ref)
    p = r.params(w, h, samples=2, max_depth=8, seed=9)
    side = torch.cuda.Stream()
    out = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    with torch.cuda.stream(side):
        r.render_device(p, out.data_ptr(), w * h, side.cuda_stream)       # outside the capture: buffers of this size exist afterwards
    side.synchronize()
    assert r.stats()["wavefront_items"] >= 2 * w * h
    assert np.array_equal(out.cpu().numpy().view(np.uint32).reshape(h, w), ref)
    graph = torch.cuda.CUDAGraph()
    out.zero_()
    with torch.cuda.graph(graph, stream=side):
        r.render_device(p, out.data_ptr(), w * h, side.cuda_stream)
    assert int(out.abs().sum().item()) == 0        # recorded, not run
    for _ in range(2):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32).reshape(h, w), ref)
    again = r.render(w, h, samples=2, max_depth=8, seed=9)
    st = r.stats()
    r.close()
    assert np.array_equal(again, ref) and st["wavefront_given_up"] == 0 and st["rays_path"] % ost["rays_path"] == 0


def test_shared_queries_can_be_turned_off(gpu_renderer_factory, forced, monkeypatch):
    """The metalness and the reflectivity pass of a half-metallic surface follow one ray; the tracer walks the tree once for
    both unless SRT_NO_SHARED_QUERIES is set.  Same picture, same ray counts either way."""
    spec = deep_tree_spec()
    w, h = 96, 54
    r = gpu_renderer_factory()
    r.set_scene(*spec.host(w, h), world_kind=srt.WORLD_SURFACELIST)
    shared = r.render(w, h, samples=2, max_depth=8, seed=9)
    st_shared = r.stats()
    monkeypatch.setenv("SRT_NO_SHARED_QUERIES", "1")
    each = r.render(w, h, samples=2, max_depth=8, seed=9)
    st_each = r.stats()
    r.close()
    assert np.array_equal(shared, each)
    assert st_shared["wavefront_twins"] > 0 and st_each["wavefront_twins"] == 0
    for k in ("rays_path", "rays_shadow", "hit_nodes", "wavefront_items"):
        assert st_shared[k] == st_each[k], k
