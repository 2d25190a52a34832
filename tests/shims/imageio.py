"""Test shim for the `imageio` package imported by the reference's scripts/common.py: imread / imwrite on top of Pillow."""
import numpy as np
from PIL import Image


def imread(path):
    return np.asarray(Image.open(path))


def imwrite(path, img, **kwargs):
    img = np.asarray(img)
    if img.dtype != np.uint8:
        img = (np.clip(img, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    kw = {}
    if "quality" in kwargs:
        kw["quality"] = int(kwargs["quality"])
    im = Image.fromarray(img)
    if str(path).lower().endswith((".jpg", ".jpeg")) and im.mode == "RGBA":
        im = im.convert("RGB")
    im.save(path, **kw)
