"""The post-process filters Sorting.BlurArray / ContrastBlurArray / Despeckle (MiscFunctions.cs:122-416).

CPU: known answers for the oracle's restatement, derived by hand from the cited lines.
GPU: the library's kernels against the oracle, bit for bit (byte arithmetic in float: no tolerance).
"""
import numpy as np
import pytest

import simple_raytracer_amd as srt


def pack(r, g, b):
    return np.uint32(int(r) | (int(g) << 8) | (int(b) << 16) | 0xFF000000)


def gray_image(h, w, value):
    return np.full((h, w), pack(value, value, value), dtype=np.uint32)


def random_image(h, w, seed, bright=0.1):
    """Mostly mid-gray noise with a share of very bright and of black pixels (the ones Despeckle touches)."""
    rng = np.random.default_rng(seed)
    rgb = rng.integers(20, 180, (h, w, 3))
    hot = rng.random((h, w)) < bright
    rgb[hot] = rng.integers(230, 256, (int(hot.sum()), 3))
    rgb[rng.random((h, w)) < bright / 2] = 0
    return (rgb[..., 0] | (rgb[..., 1] << 8) | (rgb[..., 2] << 16) | 0xFF000000).astype(np.uint32)


# ---- oracle known answers ---------------------------------------------------------------------
def test_oracle_blur_uniform_picture_is_unchanged(oracle):
    img = gray_image(5, 6, 100)
    # (100 * 0.5) + ((8 * 100) / 8 * 0.5) = 100; 100 / 255 * 255 truncates back to 100?  0.39215687 * 255 = 100.00000185 -> 100
    assert (oracle.BlurArray(img, 0.5) == img).all()


def test_oracle_blur_is_in_place_in_row_major_order(oracle):
    # 3x4 picture, interior pixels (1,1) and (2,1); weight 1 -> new value = mean of the eight neighbours (MiscFunctions.cs:180-181)
    img = gray_image(3, 4, 0)
    img[1, 1] = pack(80, 80, 80)
    img[1, 2] = pack(160, 160, 160)
    out = oracle.BlurArray(img, 1.0)
    # pixel (1,1): neighbours are seven zeros and 160 -> 160 / 8 = 20
    assert out[1, 1] == pack(20, 20, 20)
    # pixel (2,1) sees the NEW value of its left neighbour (20), not 80 -> 20 / 8 = 2.5 -> (int)(2.5 / 255 * 255) = 2
    assert out[1, 2] == pack(2, 2, 2)
    # borders are never written
    assert (out[0] == img[0]).all() and (out[2] == img[2]).all() and out[1, 0] == img[1, 0] and out[1, 3] == img[1, 3]


def test_oracle_despeckle_only_touches_black_and_bright_pixels(oracle):
    img = gray_image(4, 5, 100)               # mixdown = 100 * 1.01 = 101
    img[1, 1] = pack(250, 250, 250)           # 252.5 > 200: a speckle
    img[2, 3] = pack(0, 0, 0)                 # mixdown <= 0
    out = oracle.Despeckle(img, 1.0, 200.0)
    # weight 1: blur = mean of eight = 100; new = blur (MiscFunctions.cs:401-405)
    assert out[1, 1] == pack(100, 100, 100)
    # (3,2): seven neighbours of 100 and the despeckled (1,1)... is not adjacent; eight 100s -> 100
    assert out[2, 3] == pack(100, 100, 100)
    untouched = np.ones_like(img, dtype=bool)
    untouched[1, 1] = untouched[2, 3] = False
    assert (out[untouched] == img[untouched]).all()


def test_oracle_despeckle_half_weight(oracle):
    img = gray_image(3, 3, 40)
    img[1, 1] = pack(240, 240, 240)
    out = oracle.Despeckle(img, 0.5, 200.0)
    # blur = 240 * 0.5 + 40 * 0.5 = 140; new = 240 * 0.5 + 140 * 0.5 = 190
    assert out[1, 1] == pack(190, 190, 190)


def test_oracle_contrast_blur_mixes_only_contrasting_neighbours(oracle):
    img = gray_image(3, 3, 100)
    img[1, 0] = pack(200, 200, 200)           # left: |101 - 202| > 50 -> collected first
    img[0, 1] = pack(0, 0, 0)                 # up: |101 - 0| > 50 -> differs from the first: blur = 200 / 2 + 0 / 2 = 100
    out = oracle.ContrastBlurArray(img, 1.0, 50.0)
    assert out[1, 1] == pack(100, 100, 100)
    img[0, 1] = pack(100, 100, 100)           # now only the left neighbour contrasts: blur = 200
    assert oracle.ContrastBlurArray(img, 1.0, 50.0)[1, 1] == pack(200, 200, 200)
    img[1, 0] = pack(100, 100, 100)           # nothing contrasts: blurColor = originalColor
    assert oracle.ContrastBlurArray(img, 1.0, 50.0)[1, 1] == pack(100, 100, 100)


def test_oracle_pictures_without_interior_are_unchanged(oracle):
    for shape in ((1, 1), (2, 7), (7, 2), (1, 9)):
        img = random_image(shape[0], shape[1], 3)
        assert (oracle.Despeckle(img, 1.0, 100.0) == img).all()
        assert (oracle.BlurArray(img, 1.0) == img).all()


# ---- the kernel's per-pixel arithmetic on the CPU (tests/emu runs it in the reference's row-major order) ---------
@pytest.mark.parametrize("h,w", [(3, 3), (4, 9), (37, 53), (130, 257)])
def test_kernel_source_filters_match_oracle(emu_lib, oracle, h, w):
    emu = srt.Renderer(library=emu_lib, strict=False)
    img = random_image(h, w, 11 * h + w)
    for weight, threshold in ((1.0, 200.0), (0.15, 120.0)):
        assert (emu.despeckle(img, weight, threshold) == oracle.Despeckle(img, weight, threshold)).all()
    for weight in (1.0, 0.3, 1.7, -0.5):
        assert (emu.blur_array(img, weight) == oracle.BlurArray(img, weight)).all()
    for weight, contrast in ((1.0, 40.0), (0.5, 5.0), (0.8, 0.0)):
        assert (emu.contrast_blur_array(img, weight, contrast) == oracle.ContrastBlurArray(img, weight, contrast)).all()
    twice = oracle.Despeckle(oracle.Despeckle(img, 0.8, 150.0), 0.8, 150.0)
    assert (emu.despeckle(img, 0.8, 150.0, passes=2) == twice).all()


# ---- GPU parity ------------------------------------------------------------------------------
SIZES = [(3, 3), (4, 9), (37, 53), (64, 64), (130, 257), (1080, 1920), (1100, 64), (2100, 40), (66000, 8)]   # (height, width); bands of 128 rows; more than 512 bands: two launches


@pytest.fixture(scope="module")
def gpu_filters(gpu_renderer_factory):
    r = gpu_renderer_factory()
    yield r
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", SIZES)
def test_gpu_despeckle_matches_oracle(gpu_filters, oracle, h, w):
    img = random_image(h, w, h * 7 + w)
    for weight, threshold in ((1.0, 200.0), (0.15, 120.0)):
        assert (gpu_filters.despeckle(img, weight, threshold) == oracle.Despeckle(img, weight, threshold)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", SIZES)
def test_gpu_blur_matches_oracle(gpu_filters, oracle, h, w):
    img = random_image(h, w, h + 3 * w)
    for weight in (1.0, 0.3, 1.7):
        assert (gpu_filters.blur_array(img, weight) == oracle.BlurArray(img, weight)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", SIZES)
def test_gpu_contrast_blur_matches_oracle(gpu_filters, oracle, h, w):
    img = random_image(h, w, 5 * h + w)
    for weight, contrast in ((1.0, 40.0), (0.5, 5.0)):
        assert (gpu_filters.contrast_blur_array(img, weight, contrast) == oracle.ContrastBlurArray(img, weight, contrast)).all()


@pytest.mark.gpu
def test_gpu_passes_and_degenerate_pictures(gpu_filters, oracle):
    img = random_image(90, 70, 1, bright=0.3)
    ref = img
    for _ in range(3):
        ref = oracle.Despeckle(ref, 0.8, 150.0)
    assert (gpu_filters.despeckle(img, 0.8, 150.0, passes=3) == ref).all()
    for shape in ((1, 1), (2, 7), (7, 2)):
        small = random_image(shape[0], shape[1], 9)
        assert (gpu_filters.despeckle(small, 1.0, 100.0) == small).all()
    with pytest.raises(srt.SrtError):
        gpu_filters.despeckle(img, 1.0, 100.0, passes=0)
    # the mirror of the reference's class: same names, one pass per call
    s = srt.Sorting(gpu_filters)
    assert (s.Despeckle(img.ravel(), 70, 90, 1.0, 200.0).reshape(90, 70) == oracle.Despeckle(img, 1.0, 200.0)).all()


ODD_PARAMETERS = [("blur_array", "BlurArray", (float("nan"),)), ("blur_array", "BlurArray", (-2.0,)), ("blur_array", "BlurArray", (7.5,)),
                  ("despeckle", "Despeckle", (1.0, float("nan"))), ("despeckle", "Despeckle", (float("inf"), 10.0)),
                  ("contrast_blur_array", "ContrastBlurArray", (float("inf"), -3.0)), ("contrast_blur_array", "ContrastBlurArray", (0.5, float("nan")))]


def check_odd_parameters(filters, oracle):
    """Weights and thresholds outside their ranges, NaN and infinite included, on small and ragged pictures: the
    float -> byte conversions of the recurrence have to saturate and truncate as the reference's casts do."""
    for h, w in ((3, 3), (4, 3), (3, 129), (129, 3), (40, 33)):
        img = random_image(h, w, 31 * h + w)
        for name, oracle_name, args in ODD_PARAMETERS:
            got = getattr(filters, name)(img, *args)
            assert (got == getattr(oracle, oracle_name)(img, *args)).all(), (h, w, name, args)


def test_kernel_source_takes_odd_parameters(emu_lib, oracle):
    check_odd_parameters(srt.Renderer(library=emu_lib, strict=False), oracle)


@pytest.mark.gpu
def test_gpu_takes_odd_parameters(gpu_filters, oracle):
    check_odd_parameters(gpu_filters, oracle)
