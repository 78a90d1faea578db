"""The reference's second threading technique, mixBufferTechnique (Game1.cs:56, :174-175, :209-220, :320-388)."""
import numpy as np
import pytest

import helpers
import simple_raytracer_amd as srt


def halve_into_black(img):
    """new Color(default(Color).ToVector3() / 2 + c.ToVector3() / 2) per channel, in float32 as the reference computes it."""
    out = np.zeros_like(img)
    for shift in (0, 8, 16):
        v = ((img >> shift) & 255).astype(np.float32)
        m = np.float32(0.0) * np.float32(0.5) + (v / np.float32(255.0)) * np.float32(0.5)
        out |= np.clip((m * np.float32(255.0)).astype(np.int32), 0, 255).astype(np.uint32) << shift
    return out | np.uint32(0xFF000000)


def test_oracle_one_buffer_is_the_one_sample_render_mixed_into_black(oracle):
    spec = helpers.default_spec()
    sc, cam = spec.oracle_scene(), spec.oracle_camera(64, 36)
    one, _ = sc.render(cam, 64, 36, samples=1, max_depth=4, seed=3)
    mixed, st = sc.render_mix_buffers(cam, 64, 36, 1, samples=1, max_depth=4, seed=3)
    assert (mixed == halve_into_black(one)).all()
    assert (halve_into_black(np.array([[0xFFC8C8C8]], dtype=np.uint32)) == 0xFF646464).all()   # 200 -> (int)(200/255 * 0.5 * 255) = 100


def test_oracle_divides_every_buffer_by_samples(oracle):
    """renderBlockFull adds ONE colour divided by `samples` (Game1.cs:352): with samples = 4 the picture is darker by
    sqrt(4) before the mix, whatever the number of buffers."""
    spec = helpers.default_spec()
    sc, cam = spec.oracle_scene(), spec.oracle_camera(48, 27)
    bright, _ = sc.render_mix_buffers(cam, 48, 27, 2, samples=1, max_depth=3, seed=1)
    dark, _ = sc.render_mix_buffers(cam, 48, 27, 2, samples=4, max_depth=3, seed=1)
    r_b, r_d = (bright & 255).astype(np.float64), (dark & 255).astype(np.float64)
    lit = r_b > 40
    assert abs(np.median(r_d[lit] / r_b[lit]) - 0.5) < 0.05


@pytest.mark.parametrize("n_buffers,samples,kind", [(1, 1, srt.WORLD_BVHNODE), (4, 2, srt.WORLD_SURFACELIST), (3, 500, srt.WORLD_BVHNODE)])
def test_kernel_source_mix_buffers_match_oracle(emu_lib, oracle, n_buffers, samples, kind):
    """The kernel source on the CPU (tests/emu): one-sample frames, resolve, mix -- against the oracle."""
    spec = helpers.default_spec()
    w, h = 64, 36
    emu = srt.Renderer(library=emu_lib, strict=False)
    world, lights, cam = spec.host(w, h)
    emu.set_scene(world, lights, cam, world_kind=kind)
    img = emu.render_mix_buffers(w, h, n_buffers, samples=samples, max_depth=5, seed=5)
    ref, ost = spec.oracle_scene().render_mix_buffers(spec.oracle_camera(w, h), w, h, n_buffers, samples=samples, max_depth=5, seed=5, world_kind=kind)
    assert (img == ref).all()
    assert emu.stats()["rays_path"] == ost["rays_path"]


@pytest.mark.gpu
@pytest.mark.parametrize("n_buffers,samples,kind", [(1, 1, srt.WORLD_BVHNODE), (3, 1, srt.WORLD_BVHNODE), (8, 4, srt.WORLD_SURFACELIST), (5, 500, srt.WORLD_BVHNODE)])
def test_gpu_mix_buffers_match_oracle(gpu_renderer_factory, oracle, n_buffers, samples, kind):
    spec = helpers.default_spec()
    w, h = 96, 54
    r = gpu_renderer_factory()
    world, lights, cam = spec.host(w, h)
    r.set_scene(world, lights, cam, world_kind=kind)
    img = r.render_mix_buffers(w, h, n_buffers, samples=samples, max_depth=6, seed=11)
    st = r.stats()
    ref, ost = spec.oracle_scene().render_mix_buffers(spec.oracle_camera(w, h), w, h, n_buffers, samples=samples, max_depth=6, seed=11, world_kind=kind)
    lsb, npx = helpers.max_lsb_diff(img, ref)
    assert lsb <= 1, (lsb, npx)       # the render's tolerance; in practice identical
    assert st["rays_path"] == ost["rays_path"] and st["rays_shadow"] == ost["rays_shadow"]
    with pytest.raises(srt.SrtError):
        r.render_mix_buffers(w, h, 0, samples=1)
    r.close()
