"""The one output the reference ships, testrender.jpg (README.md:10-11: a 1275 x 717 screenshot of the default scene at the
reference's own settings), against this build's renders of configuration 1 (Game1.cs:41-43, 78-79, 96-144).

tests/golden/testrender_luma_64x36.npz holds thumbnails of that picture at 64 x 36 and 320 x 180 and the position of the
point light's highlight in it (made by make_testrender_thumb.py in the build container; the reference itself does not
travel).  The screenshot is a 1275 x 717 crop of the 1280 x 720 buffer, JPEG-compressed, and was rendered with an
entropy-seeded generator, so the comparison is statistical -- but a tight one once the crop is found (every placement of
the crop inside the frame is tried; the best one is 4 columns and 1 row in):

  * the 320 x 180 thumbnails correlate above 0.9985 in every channel and differ by less than 0.0045 on average;
  * no 16 x 16 block of the thumbnail differs by more than 0.012 in its mean (3 of 255 levels);
  * the silhouettes of the two spheres in view (found by colour) start and end in the same thumbnail rows and columns;
  * the highlight of the point light has its centroid within one pixel of the reference's, and its size.

That pins what no known-answer test of a single function can: camera, scene constants, light model, the 2f + c accumulation
of the lights, recursion depth, clamping, gamma, channel order and row order of the whole path at once.  The negative controls
below show that each of these, read differently, fails at least one of the checks: one recursion level more or fewer, one
light missing, the lights summed plainly, the field of view off by a degree, a sphere off by 0.15 units, a flipped, mirrored
or channel-swapped picture."""
import os

import numpy as np
import pytest

import helpers
from golden.make_testrender_thumb import highlight, rgb_thumbnail

THUMB = os.path.join(helpers.GOLDEN, "testrender_luma_64x36.npz")
W, H, DEPTH = 1280, 720, 4          # Game1.cs:78-79, :42
MW, MH = 320, 180


def _agreement(rgb8):
    want = np.load(THUMB)["rgb"]
    got = rgb_thumbnail(rgb8)
    corr = [float(np.corrcoef(got[..., k].ravel(), want[..., k].ravel())[0, 1]) for k in range(3)]
    mad = [float(np.abs(got[..., k] - want[..., k]).mean()) for k in range(3)]
    return corr, mad


def _silhouettes(thumb):
    """Bounding boxes (first column, last column, first row, last row) of the pink metal sphere and of the yellow glossy
    sphere in a 320 x 180 thumbnail, found by colour; None for a sphere that is not there."""
    r, g, b = thumb[..., 0] * 255.0, thumb[..., 1] * 255.0, thumb[..., 2] * 255.0
    luma = 0.299 * r + 0.587 * g + 0.114 * b
    boxes = []
    for mask in (r - g > 25.0, (r - b > 25.0) & (r - g < 20.0) & (luma > 100.0)):
        cols, rows = np.nonzero(mask.sum(axis=0) >= 3)[0], np.nonzero(mask.sum(axis=1) >= 3)[0]
        boxes.append(None if len(cols) == 0 or len(rows) == 0 else (int(cols.min()), int(cols.max()), int(rows.min()), int(rows.max())))
    return boxes


def compare_with_testrender(rgb8):
    """(720, 1280, 3) uint8 render -> dict of the statistics above, at the placement of the 1275 x 717 crop that fits best."""
    fix = np.load(THUMB)
    want = fix["rgb_320x180_u16"].astype(np.float64) / 65535.0
    sw, sh = (int(v) for v in fix["source_size"])
    best = None
    for ox in range(W - sw + 1):
        for oy in range(H - sh + 1):
            got = rgb_thumbnail(rgb8[oy:oy + sh, ox:ox + sw], MW, MH).astype(np.float64)
            corr = [float(np.corrcoef(got[..., k].ravel(), want[..., k].ravel())[0, 1]) for k in range(3)]
            if best is None or np.mean(corr) > np.mean(best["corr"]):
                best = {"corr": corr, "offset": (ox, oy), "got": got}
    got, (ox, oy) = best.pop("got"), best["offset"]
    d = got - want
    best["mad"] = [float(np.abs(d[..., k]).mean()) for k in range(3)]
    best["max_block"] = float(np.abs(d[:176].reshape(11, 16, 20, 16, 3).mean(axis=(1, 3))).max())
    best["silhouettes"], best["want_silhouettes"] = _silhouettes(got), _silhouettes(want)
    best["highlight"], best["want_highlight"] = highlight(rgb8[oy:oy + sh, ox:ox + sw]), tuple(float(v) for v in fix["highlight"])
    return best


def looks_like_testrender(c):
    """The checks; returns the list of those that fail (empty: the render passes)."""
    failed = []
    if min(c["corr"]) <= 0.9985 or max(c["mad"]) >= 0.0045:
        failed.append("thumbnail")
    if c["max_block"] >= 0.012:
        failed.append("blocks")
    for got, want, slack in zip(c["silhouettes"], c["want_silhouettes"], ((1, 1, 1, 1), (1, 1, 4, 1))):   # the big sphere's top fades into the sky
        if got is None or any(abs(g - w) > s for g, w, s in zip(got, want, slack)):
            failed.append("silhouette")
    (x, y, n), (wx, wy, wn) = c["highlight"], c["want_highlight"]
    if n == 0 or abs(x - wx) > 1.0 or abs(y - wy) > 1.0 or abs(n - wn) > 0.15 * wn:
        failed.append("highlight")
    return failed


def test_thumbnail_helper_is_a_box_filter():
    flat = np.full((72, 128, 3), 51, np.uint8)
    assert np.allclose(rgb_thumbnail(flat), 0.2, atol=1e-4)
    wide = np.zeros((36, 100, 3), np.uint8)
    wide[:, 18:82] = 255                              # the centre crop to 16:9 keeps exactly these columns
    assert np.allclose(rgb_thumbnail(wide), 1.0, atol=1e-4)


def test_oracle_render_of_the_default_scene_looks_like_testrender_jpg(oracle):
    img, _ = oracle.default_scene().render(oracle.default_camera(W, H), W, H, samples=16, max_depth=DEPTH, seed=1)
    rgb = oracle.rgba_to_rgb8(img)
    corr, mad = _agreement(rgb)
    assert min(corr) > 0.998 and max(mad) < 0.006, (corr, mad)
    c = compare_with_testrender(rgb)
    assert looks_like_testrender(c) == [], c
    assert c["offset"] == (4, 1)                      # where the screenshot was cut out of the frame
    # the same picture upside down, mirrored, or with red and blue swapped
    for wrong in (rgb[::-1], rgb[:, ::-1], rgb[..., ::-1]):
        failed = looks_like_testrender(compare_with_testrender(wrong))
        assert "thumbnail" in failed and "blocks" in failed, failed


def _spec_render(oracle, spec, depth=DEPTH):
    img, _ = spec.oracle_scene().render(spec.oracle_camera(W, H), W, H, samples=4, max_depth=depth, seed=1)
    return looks_like_testrender(compare_with_testrender(oracle.rgba_to_rgb8(img)))


@pytest.mark.parametrize("control", ["depth-3", "depth-5", "no-first-light", "no-second-light", "lights-summed-plainly", "fov-64", "sphere-moved"])
def test_a_different_reading_of_the_reference_does_not_look_like_testrender_jpg(oracle, control):
    """Negative controls: each renders the default scene with one thing read differently and must fail the comparison."""
    import ctypes as C
    spec = helpers.default_spec()
    if control.startswith("depth-"):
        failed = _spec_render(oracle, spec, depth=int(control[-1]))
    elif control == "no-first-light":
        del spec.lights[0]
        failed = _spec_render(oracle, spec)
        assert "highlight" in failed
    elif control == "no-second-light":
        del spec.lights[1]
        failed = _spec_render(oracle, spec)
    elif control == "lights-summed-plainly":
        # RaytracingClasses.cs:213 read as f <- f + c instead of f <- 2 f + c (formerLightLevel += calculateLight(formerLightLevel, ...))
        fn = oracle.lib().orc_set_misreading
        fn.argtypes = [C.c_int]
        fn(1)
        try:
            failed = _spec_render(oracle, spec)
        finally:
            fn(0)
        assert "highlight" in failed
    elif control == "fov-64":
        spec.camera = (spec.camera[0], spec.camera[1], 64)       # Game1.cs:96 says 63
        failed = _spec_render(oracle, spec)
        assert "highlight" in failed
    else:
        c, r, m = spec.spheres[1]
        spec.spheres[1] = ((c[0] + 0.15, c[1], c[2]), r, m)      # the pink sphere 0.15 units to the right (Game1.cs:118)
        failed = _spec_render(oracle, spec)
        assert "silhouette" in failed
    assert "blocks" in failed, failed


@pytest.mark.gpu
def test_gpu_render_of_the_default_scene_looks_like_testrender_jpg(gpu_renderer_factory):
    import simple_raytracer_amd as srt
    from simple_raytracer_amd import game1
    r = gpu_renderer_factory()
    world, lights, cam = game1.default_scene(W, H)
    r.set_scene(world, lights, cam, world_kind=srt.WORLD_BVHNODE)
    img = r.render(W, H, samples=32, max_depth=DEPTH, seed=1)
    deeper = r.render(W, H, samples=32, max_depth=DEPTH + 1, seed=1)
    r.close()
    corr, mad = _agreement(srt.rgba_to_rgb8(img))
    assert min(corr) > 0.998 and max(mad) < 0.006, (corr, mad)
    c = compare_with_testrender(srt.rgba_to_rgb8(img))
    assert looks_like_testrender(c) == [] and c["offset"] == (4, 1), c
    assert "blocks" in looks_like_testrender(compare_with_testrender(srt.rgba_to_rgb8(deeper)))      # one recursion level more shows
