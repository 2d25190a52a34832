"""CPU: the NeRF loader's built-in image readers (host/jpeg_lite.hpp: baseline and progressive JPEG; host/testbed.cpp: PNG, every colour type / bit depth / interlacing) against the REFERENCE's decoder.
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
                ngp.read_image(p)  # (nothing is listed since round 3: the progressive fixture decodes natively, like stb_image)
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
        for trial in range(120):
            w, h = int(rng.integers(1, 70)), int(rng.integers(1, 70))
            img = rng.integers(0, 255, (h, w, 3), dtype=np.uint8) if trial % 3 else np.clip(np.add.outer(np.arange(h) * 3, np.arange(w) * 2)[..., None] + np.array([0, 40, 90]), 0, 255).astype(np.uint8)
            p = os.path.join(td, f"t{trial}.jpg")
            kw = dict(quality=int(rng.integers(20, 100)), subsampling=int(rng.integers(0, 3)), progressive=bool(trial % 2), optimize=bool(trial % 4 == 1))  # every other one progressive (SOF2)
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


def _write_png(path, samples, ctype, bits, interlace, rs, plte=None, trns=None):
    """a PNG written by hand from the published format: samples [h][w][channels] (ints), any colour type / bit depth, random filter type per row, optional Adam7"""
    import struct
    import zlib
    h, w, ch = samples.shape

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)

    def pack(row):  # [w][ch] -> bytes
        flat = row.reshape(-1)
        if bits == 16:
            return b"".join(struct.pack(">H", int(v)) for v in flat)
        if bits == 8:
            return bytes(int(v) for v in flat)
        out = bytearray((len(flat) * bits + 7) // 8)
        for i, v in enumerate(flat):
            bit = i * bits
            out[bit >> 3] |= int(v) << (8 - bits - (bit & 7))
        return bytes(out)

    def rows(sub):
        hh, ww, _ = sub.shape
        bpp = max(1, ch * bits // 8)
        out, prev = b"", None
        for y in range(hh):
            raw = pack(sub[y])
            ft = int(rs.integers(0, 5))
            f = bytearray(len(raw))
            for x in range(len(raw)):
                a = raw[x - bpp] if x >= bpp else 0
                b = prev[x] if prev is not None else 0
                c = prev[x - bpp] if (prev is not None and x >= bpp) else 0
                if ft == 0: p = 0
                elif ft == 1: p = a
                elif ft == 2: p = b
                elif ft == 3: p = (a + b) >> 1
                else:
                    q = a + b - c; pa, pb, pc = abs(q - a), abs(q - b), abs(q - c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                f[x] = (raw[x] - p) & 255
            out += bytes([ft]) + bytes(f); prev = raw
        return out
    if interlace:
        data = b""
        for xo, yo, xs, ys in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = samples[yo::ys, xo::xs]
            if sub.shape[0] and sub.shape[1]:
                data += rows(sub)
    else:
        data = rows(samples)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bits, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None:
        png += chunk(b"PLTE", bytes(plte))
    if trns is not None:
        png += chunk(b"tRNS", bytes(trns))
    z = zlib.compress(data)
    for k in range(0, len(z), 4093):  # several IDAT chunks
        png += chunk(b"IDAT", z[k:k + 4093])
    open(path, "wb").write(png + chunk(b"IEND", b""))


def test_png_variants_against_the_reference_decoder(tmp_path):
    """every colour type x bit depth x interlacing x tRNS combination of the PNG format, random filters: the host's reader returns stb_image's RGBA8 (and its 16-bit
    single-channel result, the depth-image path) bit for bit"""
    import ctypes as C
    import struct
    so = os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    sys.path.insert(0, os.path.join(ROOT, "instant-ngp_amd"))
    import pyngp as ngp
    ref = C.CDLL(so)
    rs = np.random.default_rng(5)
    n = 0
    for ctype, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))):
        chn = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
        for bits in depths:
            for interlace in (0, 1):
                for with_trns in ((False, True) if ctype in (0, 2, 3) else (False,)):
                    for (w, h) in ((1, 1), (7, 5), (33, 17)):
                        plte = trns = None
                        if ctype == 3:
                            n_pal = min(1 << bits, int(rs.integers(2, 200)))
                            plte = rs.integers(0, 256, n_pal * 3).astype(np.uint8)
                            smp = rs.integers(0, n_pal, (h, w, 1))
                            if with_trns:
                                trns = rs.integers(0, 256, int(rs.integers(1, n_pal + 1))).astype(np.uint8)
                        else:
                            smp = rs.integers(0, 1 << bits, (h, w, chn))
                            if bits == 16:
                                smp[rs.uniform(size=smp.shape) < 0.3] &= 0xff00  # so that a 16-bit colour key can match more than one pixel
                            if with_trns:
                                key = smp[int(rs.integers(0, h)), int(rs.integers(0, w))]
                                smp[rs.uniform(size=(h, w)) < 0.3] = key
                                trns = b"".join(struct.pack(">H", int(v)) for v in key)
                        f = str(tmp_path / f"v_{ctype}_{bits}_{interlace}_{int(with_trns)}_{w}x{h}.png")
                        _write_png(f, smp, ctype, bits, interlace, rs, plte, trns)
                        ww, hh = C.c_int(), C.c_int()
                        exp = np.zeros((h, w, 4), np.uint8)
                        assert ref.ref_stbi_load_rgba(f.encode(), C.byref(ww), C.byref(hh), exp.ctypes.data_as(C.POINTER(C.c_ubyte))) == 1, f
                        got = ngp.read_image(f)
                        assert got.shape == (h, w, 4) and np.array_equal(got, exp), (f, got.reshape(-1, 4)[:4].tolist(), exp.reshape(-1, 4)[:4].tolist())
                        e16 = np.zeros((h, w), np.uint16)
                        assert ref.ref_stbi_load_gray16(f.encode(), C.byref(ww), C.byref(hh), e16.ctypes.data_as(C.POINTER(C.c_ushort))) == 1
                        g16 = ngp.read_depth_png(f)
                        assert np.array_equal(np.asarray(g16).reshape(h, w), e16), f
                        n += 1
    assert n == 156


def test_jpeg_reader_survives_corruption(tmp_path):
    """baseline and progressive files with random bytes overwritten / truncated: the decoder returns pixels or refuses the file (RuntimeError from read_image), it never
    crashes (child process: a crash fails this test, not the session)"""
    import subprocess
    from PIL import Image
    rng = np.random.default_rng(9)
    img = rng.integers(0, 255, (67, 93, 3), dtype=np.uint8)
    files = []
    for k, kw in enumerate((dict(quality=85, subsampling=2), dict(quality=85, subsampling=2, progressive=True), dict(quality=40, subsampling=0, progressive=True, optimize=True))):
        p = str(tmp_path / f"f{k}.jpg"); Image.fromarray(img, "RGB").save(p, **kw); files.append(p)
    code = (f"import sys, numpy as np\nsys.path.insert(0, {os.path.join(ROOT, 'instant-ngp_amd')!r})\nimport pyngp as ngp\n"
            f"rs = np.random.default_rng(1); n_ok = n_err = 0\n"
            f"for f in {files!r}:\n"
            f"    data = bytearray(open(f, 'rb').read())\n"
            f"    for t in range(300):\n"
            f"        d = bytearray(data)\n"
            f"        for _ in range(int(rs.integers(1, 8))):\n"
            f"            d[int(rs.integers(2, len(d)))] = int(rs.integers(0, 256))\n"
            f"        if t % 5 == 0: d = d[: int(rs.integers(4, len(d)))]\n"
            f"        open(f + '.bad.jpg', 'wb').write(d)\n"
            f"        try:\n            ngp.read_image(f + '.bad.jpg'); n_ok += 1\n        except RuntimeError: n_err += 1\n"
            f"print(n_ok, n_err)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-400:])
    n_ok, n_err = map(int, r.stdout.split())
    assert n_ok + n_err == 900


def test_jpeg_scan_that_names_an_undefined_huffman_table_is_refused(tmp_path):
    """a file whose DHT segments are missing (or whose SOS names a table id no DHT defined): the tables are value-initialised and read_sos() refuses the scan -- a clean
    RuntimeError from read_image instead of decoding through uninitialised tables"""
    import pyngp as ngp
    from PIL import Image
    rng = np.random.default_rng(3)
    p = str(tmp_path / "ok.jpg")
    Image.fromarray(rng.integers(0, 255, (40, 56, 3), dtype=np.uint8), "RGB").save(p, quality=80, subsampling=2)
    data = open(p, "rb").read()
    assert ngp.read_image(p).shape[:2] == (40, 56)
    # walk the marker segments up to SOS and drop every DHT (0xFFC4)
    out, i = bytearray(data[:2]), 2
    while data[i + 1] != 0xDA:
        L = (data[i + 2] << 8) | data[i + 3]
        if data[i + 1] != 0xC4:
            out += data[i:i + 2 + L]
        i += 2 + L
    out += data[i:]
    bad = str(tmp_path / "no_dht.jpg"); open(bad, "wb").write(out)
    ngp._set_image_decoder(lambda path: None)  # no fallback decoder: the refusal must surface
    try:
        with pytest.raises(RuntimeError):
            ngp.read_image(bad)
    finally:
        ngp._set_image_decoder(ngp._pil_decoder)
