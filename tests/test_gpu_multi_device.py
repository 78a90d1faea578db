"""One context, several devices: the frame split inside libsrt.so (what the single-process C# host gets from one
srt_render call).  The test box has one GPU, so the device list names it several times: the entries are separate
replicas with their own streams and buffers, and the tiles travel by device-to-device copies (RCCL refuses one GPU in
two ranks).  With one GPU per entry the same code gathers over RCCL; srt_create checks the communicator's rank count."""
import os
import subprocess

import numpy as np
import pytest

import helpers
import simple_raytracer_amd as srt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def single(gpu_renderer_factory):
    r = gpu_renderer_factory()
    yield r
    r.close()


@pytest.mark.parametrize("n_devices", [2, 3, 8])
def test_frame_over_several_devices_is_the_single_device_frame(single, n_devices):
    w, h = 200, 120
    world, lights, cam = helpers.default_spec().host(w, h)
    single.set_scene(world, lights, cam)
    ref = single.render(w, h, samples=4, max_depth=8, seed=3)
    ref_st = single.stats()
    multi = srt.Renderer(devices=[0] * n_devices)
    info = multi.device_info()
    assert info["devices"] == [0] * n_devices and info["gather"] == srt.GATHER_PEER_COPY and info["rccl_ranks"] == 0
    multi.set_scene(world, lights, cam)
    img = multi.render(w, h, samples=4, max_depth=8, seed=3)
    st = multi.stats()
    assert np.array_equal(img, ref)
    for k in ("rays_path", "rays_shadow", "hit_nodes"):
        assert st[k] == ref_st[k], k
    parts = [multi.device_stats(i) for i in range(n_devices)]
    assert sum(p["rays_path"] for p in parts) == st["rays_path"]
    assert all(p["rays_path"] > 0 for p in parts[:min(n_devices, 4)])          # every device with tiles traced some
    assert st["kernel_ms"] == max(p["kernel_ms"] for p in parts)
    # a second frame of another size through the same context, and the counting kernels
    multi.set_scene(*helpers.default_spec().host(70, 45))
    img2 = multi.render(70, 45, samples=2, max_depth=4, seed=9, tile_w=16, tile_h=16, counted=True)
    st2 = multi.stats()
    single.set_scene(*helpers.default_spec().host(70, 45))
    ref2 = single.render(70, 45, samples=2, max_depth=4, seed=9, counted=True)
    ref2_st = single.stats()
    assert np.array_equal(img2, ref2)
    for k in ("rays_path", "box_tests", "sphere_tests"):
        assert st2[k] == ref2_st[k], k
    multi.close()


def test_async_render_over_several_devices(single):
    w, h = 160, 90
    world, lights, cam = helpers.mesh_spec().host(w, h)
    single.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
    ref = single.render(w, h, samples=2, max_depth=4, seed=5)
    multi = srt.Renderer(devices=[0, 0, 0], gather=srt.GATHER_PEER_COPY)
    multi.set_scene(world, lights, cam, world_kind=srt.WORLD_SURFACELIST)
    p = multi.params(w, h, samples=2, max_depth=4, seed=5)
    out = np.zeros(w * h, dtype=np.uint32)
    multi.render_async(p, out)
    while not multi.poll():
        pass
    multi.wait()
    assert np.array_equal(out.reshape(h, w), ref)
    assert multi.stats()["rays_path"] == single.stats()["rays_path"]
    # entry points that run on the first device alone still work in such a context
    assert np.array_equal(multi.render_mix_buffers(w, h, 3, samples=4, max_depth=3, seed=2),
                          single.render_mix_buffers(w, h, 3, samples=4, max_depth=3, seed=2))
    assert np.array_equal(multi.despeckle(ref, 1.0, 120.0, passes=2), single.despeckle(ref, 1.0, 120.0, passes=2))
    with pytest.raises(srt.SrtError):
        multi.render(w, h, samples=1, layout=srt.LAYOUT_TILES, rank=1, world_size=2)
    multi.close()


def test_a_device_list_of_one_with_rccl_requested_needs_no_communicator():
    r = srt.Renderer(devices=[0], gather=srt.GATHER_RCCL)
    assert r.device_info() == {"devices": [0], "gather": srt.GATHER_PEER_COPY, "rccl_ranks": 0}
    r.close()
    r = srt.Renderer(devices="all")
    assert len(r.device_info()["devices"]) >= 1
    r.close()
    with pytest.raises(srt.SrtError):
        srt.Renderer(devices=[99])


def test_rccl_loads_and_makes_a_communicator(monkeypatch):
    """The RCCL side of a several-device context on the one GPU of this box: librccl is found and bound at run time,
    ncclCommInitAll makes a communicator and its rank count is the device count.  (Sends and receives need a second GPU.)"""
    monkeypatch.setenv("SRT_RCCL_SINGLE", "1")
    r = srt.Renderer(devices=[0], gather=srt.GATHER_RCCL)
    assert r.device_info() == {"devices": [0], "gather": srt.GATHER_RCCL, "rccl_ranks": 1}
    world, lights, cam = helpers.default_spec().host(64, 36)
    r.set_scene(world, lights, cam)
    img = r.render(64, 36, samples=1, max_depth=4, seed=1)
    assert img.shape == (36, 64) and (img >> 24 == 255).all()
    r.close()


def test_launches_on_different_streams_are_ordered(single):
    """Launches of one context share its sample buffer and work counter: a launch on another stream has to wait for
    the one before it, or both pictures are wrong."""
    import torch
    w, h = 320, 180
    world, lights, cam = helpers.default_spec().host(w, h)
    single.set_scene(world, lights, cam)
    want = [single.render(w, h, samples=8, max_depth=8, seed=s) for s in (1, 2, 3)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [torch.zeros(w * h, dtype=torch.int32, device="cuda") for _ in range(3)]
    for i, s in enumerate((1, 2, 3)):
        p = single.params(w, h, samples=8, max_depth=8, seed=s)
        single.render_device(p, outs[i].data_ptr(), outs[i].numel(), streams[i].cuda_stream)
    st = single.stats()      # waits for the last launch, which waited for the others
    torch.cuda.synchronize()
    for i in range(3):
        assert np.array_equal(outs[i].cpu().numpy().view(np.uint32).reshape(h, w), want[i]), i
    assert st["rays_path"] > 0


def test_c_host_over_several_devices_writes_the_same_frame(tmp_path):
    exe = os.path.join(helpers.ROOT, "examples", "game1_host")
    frames = []
    for devices in ("0", "-3"):
        out = tmp_path / ("frame%s.ppm" % devices)
        subprocess.run([exe, "160", "90", "3", "4", "7", str(out), "0", devices], check=True, timeout=100)
        frames.append(out.read_bytes())
    assert frames[0] == frames[1]


@pytest.mark.parametrize("n_devices,n_buffers", [(2, 5), (3, 3), (3, 8), (8, 4)])
def test_mix_buffers_over_several_devices_is_the_single_device_picture(single, n_devices, n_buffers):
    """The mix-buffer technique (Game1.cs:174-175, 209-220): every device renders whole one-sample frames, the first device
    mixes them in buffer order -- the picture and the ray counts are those of one device."""
    w, h = 160, 90
    world, lights, cam = helpers.default_spec().host(w, h)
    single.set_scene(world, lights, cam)
    ref = single.render_mix_buffers(w, h, n_buffers, samples=2, max_depth=6, seed=4)
    ref_st = single.stats()
    multi = srt.Renderer(devices=[0] * n_devices)
    multi.set_scene(world, lights, cam)
    img = multi.render_mix_buffers(w, h, n_buffers, samples=2, max_depth=6, seed=4)
    st = multi.stats()
    assert np.array_equal(img, ref)
    assert st["rays_path"] == ref_st["rays_path"] and st["rays_shadow"] == ref_st["rays_shadow"]
    parts = [multi.device_stats(i)["rays_path"] for i in range(n_devices)]
    assert sum(parts) == st["rays_path"] and all(p > 0 for p in parts[:min(n_devices, n_buffers)])
    # and again through the same context (the buffers of the first render are reused)
    assert np.array_equal(multi.render_mix_buffers(w, h, n_buffers, samples=2, max_depth=6, seed=4), ref)
    multi.close()


@pytest.mark.parametrize("n_devices", [2, 3])
def test_progressive_render_over_several_devices(single, n_devices):
    """The picture that fills in while it renders (Game1.cs:188-201), with the frame split over the devices of the context:
    after every pass a complete preview, at the end the frame one device renders."""
    w, h, spp = 200, 120, 12
    world, lights, cam = helpers.default_spec().host(w, h)
    single.set_scene(world, lights, cam)
    want = single.render(w, h, samples=spp, max_depth=6, seed=3)
    want_st = single.stats()
    multi = srt.Renderer(devices=[0] * n_devices)
    multi.set_scene(world, lights, cam)
    p = multi.params(w, h, samples=spp, max_depth=6, seed=3)
    out, preview = np.zeros(w * h, dtype=np.uint32), np.zeros(w * h, dtype=np.uint32)
    multi.render_progressive(p, out, 4)
    seen, polls = set(), 0
    while True:
        done, total = multi.progress()
        assert total == 4 and 0 <= done <= 4
        if done >= 1 and done not in seen:
            multi.read_preview(preview)
            seen.add(done)
            assert (preview >> 24 == 255).all()           # a complete picture: every tile of every device has arrived
        if multi.poll():
            break
        polls += 1
        assert polls < 10_000_000
    assert np.array_equal(out.reshape(h, w), want)
    st = multi.stats()
    assert st["rays_path"] == want_st["rays_path"] and st["rays_shadow"] == want_st["rays_shadow"]
    # the context renders as usual afterwards
    assert np.array_equal(multi.render(w, h, samples=spp, max_depth=6, seed=3), want)
    assert multi.progress() == (0, 0)
    multi.close()
