"""CPU: the NeRF loader's built-in image readers (host/jpeg_lite.hpp: baseline JPEG; host/testbed.cpp: PNG 8 / 16 bit) against the REFERENCE's decoder.
The reference loads its training images with the vendored stb_image (load_stbi / load_stbi_16, nerf_loader.cu:570-603, 633); tools/make_image_golden.py
compiled that decoder from the reference's tree (oracle/_ref) and recorded sha256 of its output for the fixtures under tests/golden/images/ and for all 50
frames of data/nerf/fox.  Bar: bit-exact (the decoded bytes are the training pixels)."""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD_DIR = os.path.join(ROOT, "tests", "golden", "images")
sys.path.insert(0, os.path.join(ROOT, "instant-ngp_amd"))


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(GOLD_DIR, "golden.json")))


def test_fixtures_decode_like_the_reference(gold):
    import pyngp as ngp
    for name, g in gold["rgba8"].items():
        p = os.path.join(GOLD_DIR, name)
        if name in gold["not_decoded_by_jpeg_lite"]:
            with pytest.raises(RuntimeError):
                ngp.read_image(p)  # progressive JPEG: left to the decoder hook (Pillow), never decoded wrongly
            continue
        a = ngp.read_image(p)
        assert list(a.shape) == g["shape"], name
        assert hashlib.sha256(a.tobytes()).hexdigest() == g["sha256"], f"{name}: decoded pixels differ from stb_image's"
    for name, g in gold["gray16"].items():
        a = ngp.read_depth_png(os.path.join(GOLD_DIR, name))
        assert list(a.shape) == g["shape"] and hashlib.sha256(a.tobytes()).hexdigest() == g["sha256"], f"{name}: 16-bit channel differs from stbi_load_16's"


def test_fox_frames_decode_like_the_reference(gold):
    """all 50 JPEGs (1080 x 1920, 4:2:0) of the reference's shipped capture, natively in C++ -- a C++ caller of Testbed::load_training_data needs no Pillow"""
    import pyngp as ngp
    d = os.path.join(ROOT, "_ref_data", "data", "nerf", "fox", "images")
    if not os.path.isdir(d):
        pytest.skip("_ref_data/ not staged")
    assert len(gold["fox"]) == 50
    for name, g in gold["fox"].items():
        a = ngp.read_image(os.path.join(d, name))
        assert list(a.shape) == g["shape"] and hashlib.sha256(a.tobytes()).hexdigest() == g["sha256"], name


def test_against_the_reference_decoder_directly(gold):
    """with oracle/_ref present (built from /root/reference by oracle/Makefile): random JPEGs of many shapes and qualities, pixel for pixel"""
    from PIL import Image
    import pyngp as ngp
    so = os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    ref = C.CDLL(so)
    rng = np.random.default_rng(3)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        for trial in range(40):
            w, h = int(rng.integers(1, 70)), int(rng.integers(1, 70))
            img = rng.integers(0, 255, (h, w, 3), dtype=np.uint8) if trial % 3 else np.clip(np.add.outer(np.arange(h) * 3, np.arange(w) * 2)[..., None] + np.array([0, 40, 90]), 0, 255).astype(np.uint8)
            p = os.path.join(td, f"t{trial}.jpg")
            kw = dict(quality=int(rng.integers(20, 100)), subsampling=int(rng.integers(0, 3)))
            (Image.fromarray(img, "RGB") if trial % 5 else Image.fromarray(img[..., 0], "L")).save(p, **kw)
            a = ngp.read_image(p)
            ww, hh = C.c_int(), C.c_int()
            b = np.empty((h, w, 4), np.uint8)
            assert ref.ref_stbi_load_rgba(p.encode(), C.byref(ww), C.byref(hh), b.ctypes.data_as(C.c_void_p)) == 1
            assert (ww.value, hh.value) == (w, h) and np.array_equal(a, b), (trial, w, h, kw, int(np.abs(a.astype(int) - b).max()))


def test_fox_loads_without_the_python_decoder():
    """Testbed.load_training_data on the shipped capture with the Pillow hook removed: the C++ host decodes the JPEGs itself"""
    import pyngp as ngp
    scene = os.path.join(ROOT, "_ref_data", "data", "nerf", "fox", "transforms.json")
    if not os.path.exists(scene):
        pytest.skip("_ref_data/ not staged")
    ngp._set_image_decoder(lambda path: None)  # a hook that decodes nothing
    try:
        t = ngp.Testbed()
        t.load_training_data(scene)
        assert t.nerf.training.dataset.n_images == 50 and t.nerf.training.dataset.metadata[0].resolution == [1080, 1920]
    finally:
        ngp._set_image_decoder(ngp._pil_decoder)
