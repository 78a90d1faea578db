"""Makes tests/golden/testrender_luma_64x36.npz from the only output the reference ships: testrender.jpg (README.md:10-11),
a 1275 x 717 screenshot of the default scene.  Run in the build container (it reads /root/reference); the fixture is data:
thumbnails of the picture, centre-cropped to 16:9, at 64 x 36 and at 320 x 180 (16 bits per channel), and where the
point light's highlight sits in the full-size picture (centroid and pixel count above a luminance threshold).

    python tests/golden/make_testrender_thumb.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/testrender.jpg"


def _shrink(plane, width, height):
    img = Image.fromarray((plane * 65535.0).astype(np.uint16).astype(np.int32), mode="I")
    return (np.asarray(img.resize((width, height), Image.BOX), dtype=np.float64) / 65535.0).astype(np.float32)


def rgb_thumbnail(rgb, width=64, height=36):
    """(H, W, 3) uint8 -> (height, width, 3) float32 in 0..1: centre crop to width:height, then box filter."""
    rgb = np.asarray(rgb, dtype=np.float64) / 255.0
    h, w = rgb.shape[:2]
    if w * height > h * width:                     # too wide
        cw = h * width // height
        x0 = (w - cw) // 2
        rgb = rgb[:, x0:x0 + cw]
    else:
        ch = w * height // width
        y0 = (h - ch) // 2
        rgb = rgb[y0:y0 + ch]
    return np.stack([_shrink(rgb[..., k], width, height) for k in range(3)], axis=-1)


def luma_thumbnail(rgb, width=64, height=36):
    """(H, W, 3) uint8 -> (height, width) float32 luminance in 0..1 of the thumbnail."""
    t = rgb_thumbnail(rgb, width, height)
    return (0.299 * t[..., 0] + 0.587 * t[..., 1] + 0.114 * t[..., 2]).astype(np.float32)


HIGHLIGHT_LUMA = 235.0      # of 255: only the point light's specular highlight on the big sphere is this bright


def highlight(rgb):
    """(H, W, 3) uint8 -> (x, y, pixels): centroid and size of what is brighter than HIGHLIGHT_LUMA; (-1, -1, 0) if nothing is."""
    a = np.asarray(rgb, dtype=np.float64)
    ys, xs = np.nonzero(0.299 * a[..., 0] + 0.587 * a[..., 1] + 0.114 * a[..., 2] > HIGHLIGHT_LUMA)
    return (float(xs.mean()), float(ys.mean()), int(len(xs))) if len(xs) else (-1.0, -1.0, 0)


if __name__ == "__main__":
    rgb = np.asarray(Image.open(SRC).convert("RGB"))
    thumb = luma_thumbnail(rgb)
    mid = np.round(rgb_thumbnail(rgb, 320, 180) * 65535.0).astype(np.uint16)
    np.savez_compressed(os.path.join(HERE, "testrender_luma_64x36.npz"), luma=thumb, rgb=rgb_thumbnail(rgb),
                        source_size=np.array(rgb.shape[:2][::-1]), rgb_320x180_u16=mid, highlight=np.array(highlight(rgb)))
    print("wrote thumbnail", thumb.shape, "of", rgb.shape, "mean", float(thumb.mean()))
