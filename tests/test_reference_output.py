"""The one output the reference ships, testrender.jpg (README.md:10-11: a 1275 x 717 screenshot of the default scene at the
reference's own settings), against this build's renders of configuration 1 (Game1.cs:41-43, 78-79, 96-144).

tests/golden/testrender_luma_64x36.npz holds a 64 x 36 thumbnail of that picture (made by make_testrender_thumb.py in
the build container; the reference itself does not travel).  The screenshot is cropped by a few pixels, JPEG-compressed
and was rendered with an entropy-seeded generator, so the comparison is statistical: thumbnails of the two pictures must
correlate above 0.998 in every channel and differ by less than 0.006 (1.5 of 255 levels) on average.  That pins what no
known-answer test of a single function can: camera, scene constants, light model, the 2f + c accumulation, clamping,
gamma, channel order and row order of the whole path at once -- a wrong sign or a swapped channel anywhere drops the
correlation far below the bound."""
import os

import numpy as np
import pytest

import helpers
from golden.make_testrender_thumb import rgb_thumbnail

THUMB = os.path.join(helpers.GOLDEN, "testrender_luma_64x36.npz")
W, H, DEPTH = 1280, 720, 4          # Game1.cs:78-79, :42


def _agreement(rgb8):
    want = np.load(THUMB)["rgb"]
    got = rgb_thumbnail(rgb8)
    corr = [float(np.corrcoef(got[..., k].ravel(), want[..., k].ravel())[0, 1]) for k in range(3)]
    mad = [float(np.abs(got[..., k] - want[..., k]).mean()) for k in range(3)]
    return corr, mad


def test_thumbnail_helper_is_a_box_filter():
    flat = np.full((72, 128, 3), 51, np.uint8)
    assert np.allclose(rgb_thumbnail(flat), 0.2, atol=1e-4)
    wide = np.zeros((36, 100, 3), np.uint8)
    wide[:, 18:82] = 255                              # the centre crop to 16:9 keeps exactly these columns
    assert np.allclose(rgb_thumbnail(wide), 1.0, atol=1e-4)


def test_oracle_render_of_the_default_scene_looks_like_testrender_jpg(oracle):
    img, _ = oracle.default_scene().render(oracle.default_camera(W, H), W, H, samples=4, max_depth=DEPTH, seed=1)
    corr, mad = _agreement(oracle.rgba_to_rgb8(img))
    assert min(corr) > 0.998 and max(mad) < 0.006, (corr, mad)
    # the comparison has teeth: the same picture upside down, or with red and blue swapped, fails it
    corr_flipped, _ = _agreement(oracle.rgba_to_rgb8(img)[::-1])
    corr_bgr, mad_bgr = _agreement(oracle.rgba_to_rgb8(img)[..., ::-1])
    assert min(corr_flipped) < 0.5 and (min(corr_bgr) < 0.99 or max(mad_bgr) > 0.01)


@pytest.mark.gpu
def test_gpu_render_of_the_default_scene_looks_like_testrender_jpg(gpu_renderer_factory):
    import simple_raytracer_amd as srt
    from simple_raytracer_amd import game1
    r = gpu_renderer_factory()
    world, lights, cam = game1.default_scene(W, H)
    r.set_scene(world, lights, cam, world_kind=srt.WORLD_BVHNODE)
    img = r.render(W, H, samples=32, max_depth=DEPTH, seed=1)
    r.close()
    corr, mad = _agreement(srt.rgba_to_rgb8(img))
    assert min(corr) > 0.998 and max(mad) < 0.006, (corr, mad)
