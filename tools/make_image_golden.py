#!/usr/bin/env python
"""Golden vectors for the loader's built-in image readers (host/jpeg_lite.hpp, the PNG reader of host/testbed.cpp), generated with the REFERENCE's decoder:
stb_image compiled from /root/reference/dependencies/stb_image by oracle/Makefile (oracle/_ref/libstb_ref.so).  Writes a handful of small fixture files
(JPEG: 4:4:4 / 4:2:2 / 4:2:0 / grayscale / odd sizes / restart intervals, written by Pillow; PNG: 8 / 16 bit) under tests/golden/images/ and
tests/golden/images/golden.json = sha256 of the reference's decoded RGBA8 (or 16-bit gray) bytes per file, plus the hashes of the first fox frames.
Run here (the reference tree is needed); tests/test_jpeg.py checks the committed fixtures without it."""
import ctypes as C
import glob
import hashlib
import json
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "images")
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so"))


def ref_rgba(path):
    w, h = C.c_int(), C.c_int()
    assert ref.ref_stbi_load_rgba(path.encode(), C.byref(w), C.byref(h), None) == 1, path
    out = np.empty((h.value, w.value, 4), np.uint8)
    ref.ref_stbi_load_rgba(path.encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p))
    return out


def ref_gray16(path):
    w, h = C.c_int(), C.c_int()
    assert ref.ref_stbi_load_gray16(path.encode(), C.byref(w), C.byref(h), None) == 1, path
    out = np.empty((h.value, w.value), np.uint16)
    ref.ref_stbi_load_gray16(path.encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(12)
    yy, xx = np.mgrid[0:61, 0:83]
    img = np.stack([127 + 120 * np.sin(xx / 7.0) * np.cos(yy / 5.0), 127 + 120 * np.cos(xx / 11.0 + yy / 3.0), (xx * 3 + yy * 2) % 256], -1)
    img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
    pil = Image.fromarray(img, "RGB")
    pil.save(os.path.join(OUT, "rgb_444_q90.jpg"), quality=90, subsampling=0)
    pil.save(os.path.join(OUT, "rgb_422_q75.jpg"), quality=75, subsampling=1)
    pil.save(os.path.join(OUT, "rgb_420_q60.jpg"), quality=60, subsampling=2)
    pil.save(os.path.join(OUT, "rgb_420_q95_optimized.jpg"), quality=95, subsampling=2, optimize=True)
    pil.convert("L").save(os.path.join(OUT, "gray_q80.jpg"), quality=80)
    pil.crop((0, 0, 17, 9)).save(os.path.join(OUT, "rgb_420_17x9.jpg"), quality=85, subsampling=2)
    pil.crop((0, 0, 1, 1)).save(os.path.join(OUT, "rgb_420_1x1.jpg"), quality=85, subsampling=2)
    try:
        pil.save(os.path.join(OUT, "rgb_420_restart.jpg"), quality=80, subsampling=2, restart_marker_blocks=3)
    except TypeError:
        pass
    pil.save(os.path.join(OUT, "progressive_not_supported.jpg"), quality=80, progressive=True)
    rgba = np.concatenate([img, ((xx + yy) % 256).astype(np.uint8)[..., None]], -1)
    Image.fromarray(rgba, "RGBA").save(os.path.join(OUT, "rgba8.png"))
    Image.fromarray(img, "RGB").save(os.path.join(OUT, "rgb8.png"))
    Image.fromarray(img[..., 0], "L").save(os.path.join(OUT, "gray8.png"))
    d16 = (rng.integers(0, 65535, (61, 83))).astype(np.uint16)
    Image.fromarray(d16, "I;16").save(os.path.join(OUT, "gray16.png"))
    gold = {"generator": "tools/make_image_golden.py with stb_image from /root/reference/dependencies/stb_image (oracle/_ref)", "rgba8": {}, "gray16": {}, "not_decoded_by_jpeg_lite": [], "fox": {}}
    for f in sorted(os.listdir(OUT)):
        p = os.path.join(OUT, f)
        if f.endswith((".jpg", ".png")):
            a = ref_rgba(p)
            gold["rgba8"][f] = {"shape": list(a.shape), "sha256": hashlib.sha256(a.tobytes()).hexdigest()}
        if f.endswith(".png"):
            g = ref_gray16(p)
            gold["gray16"][f] = {"shape": list(g.shape), "sha256": hashlib.sha256(g.tobytes()).hexdigest()}
    for p in sorted(glob.glob("/root/reference/data/nerf/fox/images/*.jpg")):
        a = ref_rgba(p)
        gold["fox"][os.path.basename(p)] = {"shape": list(a.shape), "sha256": hashlib.sha256(a.tobytes()).hexdigest()}
    json.dump(gold, open(os.path.join(OUT, "golden.json"), "w"), indent=1)
    print(len(gold["rgba8"]), "fixtures,", len(gold["fox"]), "fox frames")


if __name__ == "__main__":
    main()
