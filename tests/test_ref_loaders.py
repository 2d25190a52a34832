"""CPU: the host's data-format readers against the reference's OWN dependencies, compiled from /root/reference where they lie into oracle/_ref/libloaders_ref.so
(oracle/ref_loaders_wrapper.cpp; test infrastructure, git-ignored, built by oracle/Makefile when the reference tree is present):
  host/exr_lite.hpp   vs tinyexr's LoadEXR (what src/tinyexr_wrapper.cu calls)                    -- every pixel, bit for bit
  the frame order     vs NaturalSort's SI::natural::compare (nerf_loader.cu:347-349)              -- the same strict weak order on awkward names
  host/mesh_lite.hpp  vs tinyobjloader's LoadObj as src/tinyobj_loader_wrapper.cu drives it       -- every triangle, bit for bit
Skipped when oracle/_ref is not built (a checkout without the reference tree)."""
import ctypes as C
import itertools
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instant-ngp_amd"))
SO = os.path.join(ROOT, "oracle", "_ref", "libloaders_ref.so")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(SO):
        pytest.skip("oracle/_ref/libloaders_ref.so not built (needs /root/reference; `make -C oracle ref`)")
    lib = C.CDLL(SO)
    lib.ref_obj_load_triangles.restype = C.c_longlong
    return lib


@pytest.fixture(scope="module")
def ngp():
    import pyngp
    return pyngp


def _ref_exr(ref, path):
    w, h = C.c_int(), C.c_int()
    assert ref.ref_exr_load_rgba(path.encode(), C.byref(w), C.byref(h), None) == 1, path
    out = np.zeros((h.value, w.value, 4), np.float32)
    assert ref.ref_exr_load_rgba(path.encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.POINTER(C.c_float))) == 1
    return out


def test_exr_reader_equals_tinyexr_on_albert(ref, ngp):
    src = os.path.join(ROOT, "_ref_data", "data", "image", "albert.exr")
    if not os.path.exists(src):
        pytest.skip("_ref_data/ not staged")
    a, b = ngp.read_exr(src), _ref_exr(ref, src)
    assert a.shape == b.shape == (1024, 1024, 4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("compression", [0, 2, 3])  # NO, ZIPS (one scan line per block), ZIP (16 lines per block)
@pytest.mark.parametrize("pixel_type", ["float", "half"])
def test_exr_reader_equals_tinyexr_on_written_files(ref, ngp, tmp_path, compression, pixel_type):
    """scan-line files written here by the published layout (RGBA, channels in file order A B G R, odd sizes, values incl. negatives / huge / denormals):
    both readers must return the same floats"""
    import struct
    import zlib
    w, h = 37, 21
    rs = np.random.default_rng(7)
    px = rs.normal(0, 3, (h, w, 4)).astype(np.float32)
    px[0, 0] = [0.0, -0.0, 65504.0, 1e-7]; px[1, 1] = [1e4, -1e4, 6e-8, 1.0]
    if pixel_type == "half":
        px = px.astype(np.float16).astype(np.float32)
    tcode, dt = (2, np.float32) if pixel_type == "float" else (1, np.float16)

    def attr(name, typ, payload):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", tcode, 0, 1, 1) for n in ("A", "B", "G", "R")) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    head = struct.pack("<II", 20000630, 2) + attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([compression])) + \
        attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + \
        attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + \
        attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines_per_block = 16 if compression == 3 else 1

    def zip_block(raw):  # OpenEXR ZIP: de-interleave even / odd bytes, delta-predict, deflate; stored raw when that is not smaller
        t = bytes(raw[0::2]) + bytes(raw[1::2])
        d = bytearray(len(t)); d[0] = t[0]
        for i in range(1, len(t)):
            d[i] = (t[i] - t[i - 1] + 128) & 255
        z = zlib.compress(bytes(d))
        return z if len(z) < len(raw) else raw
    blocks = []
    for y0 in range(0, h, lines_per_block):
        raw = b"".join(px[y, :, k].astype(dt).tobytes() for y in range(y0, min(y0 + lines_per_block, h)) for k in (3, 2, 1, 0))
        data = raw if compression == 0 else zip_block(raw)
        blocks.append(struct.pack("<ii", y0, len(data)) + data)
    offs, pos = [], len(head) + 8 * len(blocks)
    for b in blocks:
        offs.append(pos); pos += len(b)
    f = tmp_path / f"t_{compression}_{pixel_type}.exr"
    f.write_bytes(head + struct.pack(f"<{len(blocks)}Q", *offs) + b"".join(blocks))
    a, b = ngp.read_exr(str(f)), _ref_exr(ref, str(f))
    assert a.shape == b.shape == (h, w, 4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), np.abs(a - b).max()
    assert np.array_equal(a.view(np.uint32), px.view(np.uint32))  # and both return what was written


def test_frame_order_equals_naturalsort(ref, ngp):
    names = ["r_0.png", "r_1.png", "r_2.png", "r_10.png", "r_010.png", "r_9.png", "r_09.png", "r_100.png", "r_1a.png", "r_1b2.png", "r_1b10.png", "images/0001.jpg",
             "images/0002.jpg", "images/0010.jpg", "images/2.jpg", "images/10", "images/10.png", "a", "a0", "a00", "a1", "b", "", "0", "00", "1", "01", "10", "2", "A1", "a_1", "a-1",
             "frame_1_2", "frame_1_10", "frame_01_3", "x9y8", "x9y10", "x10y1", "./train/r_7", "./train/r_70", "./train/r_8"]
    for a, b in itertools.product(names, repeat=2):
        assert bool(ref.ref_natural_less(a.encode(), b.encode())) == ngp._natural_less(a, b), (a, b)
    # the dataset's own frames: sorted the same way
    import json
    tf = os.path.join(ROOT, "_ref_data", "data", "nerf", "fox", "transforms.json")
    if os.path.exists(tf):
        paths = [f["file_path"] for f in json.load(open(tf))["frames"]]
        import functools
        cmp_ref = functools.cmp_to_key(lambda a, b: -1 if ref.ref_natural_less(a.encode(), b.encode()) else (1 if ref.ref_natural_less(b.encode(), a.encode()) else 0))
        cmp_own = functools.cmp_to_key(lambda a, b: -1 if ngp._natural_less(a, b) else (1 if ngp._natural_less(b, a) else 0))
        assert sorted(paths, key=cmp_ref) == sorted(paths, key=cmp_own)


def _ref_obj(ref, path):
    n = ref.ref_obj_load_triangles(path.encode(), None, C.c_longlong(0))
    assert n > 0 and n % 9 == 0, n
    out = np.zeros(n, np.float32)
    assert ref.ref_obj_load_triangles(path.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), C.c_longlong(n)) == n
    return out.reshape(-1, 3, 3)


def test_obj_reader_equals_tinyobjloader_on_armadillo(ref, ngp):
    src = os.path.join(ROOT, "_ref_data", "data", "sdf", "armadillo.obj")
    if not os.path.exists(src):
        pytest.skip("_ref_data/ not staged")
    a, b = ngp.read_obj(src), _ref_obj(ref, src)
    assert a.shape == b.shape and a.shape[0] == 99976
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_obj_reader_equals_tinyobjloader_on_awkward_file(ref, ngp, tmp_path):
    """comments, texture / normal indices, negative (relative) indices, quads (tinyobjloader splits them along the shorter diagonal: both cases and the tie), exponents,
    CRLF line ends, several groups / objects.  (Faces with five and more corners: test_obj_polygons_are_ear_clipped_like_tinyobjloader.)"""
    txt = "\r\n".join([
        "# a comment", "o first", "v 0 0 0", "v 1 0 0", "v 1 1 0", "v 0 1 0", "v 0.5 0.5 1e0", "v -2.5e-1 3.25 +4", "vt 0 0", "vt 1 1", "vn 0 0 1",
        "g tri", "f 1 2 3", "f 1/1 3/2 4/1", "f 1//1 2//1 5//1", "f 2/1/1 3/2/1 5/1/1",
        "g quad", "f 1 2 3 4", "f -6 -5 -4 -3", "f 1 2 6 4", "f 2 6 4 1", "f 5 6 1 2", "o second", "f 6 5 4", ""])
    f = tmp_path / "awkward.obj"; f.write_text(txt)
    a, b = ngp.read_obj(str(f)), _ref_obj(ref, str(f))
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _write_with_reference(w, h, compression, half, path, seed=0):
    """The reference's own tinyexr ENCODER writes the file (in a child process: its PIZ encoder corrupts the heap on very small images)."""
    import subprocess
    code = (f"import ctypes as C, numpy as np\n"
            f"ref = C.CDLL({SO!r})\n"
            f"px = np.random.default_rng({seed}).normal(0, 3, ({h}, {w}, 4)).astype(np.float32); px[..., 3] = 1.0; px[0, 0, :3] = [65504.0, -0.0, 1e-7]\n"
            f"np.save({path + '.npy'!r}, px)\n"
            f"assert ref.ref_exr_save_rgba({path!r}.encode(), {w}, {h}, px.ctypes.data_as(C.POINTER(C.c_float)), {compression}, {int(half)}) == 1\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    return r.returncode == 0 and os.path.exists(path)


@pytest.mark.parametrize("compression", [1, 4])  # RLE, PIZ
@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("size", [(37, 45), (64, 32), (600, 70), (33, 33), (129, 97)])
def test_exr_rle_and_piz_written_by_tinyexr(ref, ngp, tmp_path, compression, half, size):
    """RLE and PIZ (wavelet + Huffman; 32 scan lines per block; the 600-wide float case has > 2^14 distinct words per block = the 16-bit wavelet, the small ones the
    14-bit one) files written by the reference's encoder: the host reader, tinyexr's reader and the written data agree bit for bit"""
    w, h = size
    f = str(tmp_path / f"t_{w}x{h}_{compression}_{int(half)}.exr")
    if not _write_with_reference(w, h, compression, half, f):
        pytest.skip("the reference's tinyexr encoder failed on this shape")
    px = np.load(f + ".npy")
    exp = px.astype(np.float16).astype(np.float32) if half else px
    a, b = ngp.read_exr(f), _ref_exr(ref, f)
    assert a.shape == b.shape == (h, w, 4)
    if not half:  # (the encoder's float -> half conversion is its own business: it rounds a few subnormals differently from numpy)
        assert np.array_equal(b.view(np.uint32), exp.view(np.uint32))
    else:
        assert np.allclose(b, exp, rtol=2e-3, atol=1e-7)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (np.abs(a - b).max(), int((a != b).sum()))


def test_exr_piz_reader_survives_corruption(ref, tmp_path):
    """every size, code length and offset in a PIZ block comes from the file: 400 random corruptions of block bytes must end in RuntimeError or a decoded image,
    never in a crash (run in a child process so that a crash is a failure of this test, not of the session)"""
    import subprocess
    f = str(tmp_path / "p.exr")
    if not _write_with_reference(96, 40, 4, True, f):
        pytest.skip("the reference's tinyexr encoder failed")
    code = (f"import sys, numpy as np\nsys.path.insert(0, {os.path.join(ROOT, 'instant-ngp_amd')!r})\nimport pyngp as ngp\n"
            f"data = bytearray(open({f!r}, 'rb').read()); rs = np.random.default_rng(3); n_ok = n_err = 0\n"
            f"start = len(data) // 6\n"
            f"for t in range(400):\n"
            f"    d = bytearray(data)\n"
            f"    for _ in range(int(rs.integers(1, 6))):\n"
            f"        d[int(rs.integers(start, len(d)))] = int(rs.integers(0, 256))\n"
            f"    if t % 7 == 0: d = d[: int(rs.integers(start, len(d)))]\n"
            f"    open({f + '.bad'!r}, 'wb').write(d)\n"
            f"    try:\n        ngp.read_exr({f + '.bad'!r}); n_ok += 1\n    except RuntimeError: n_err += 1\n"
            f"print(n_ok, n_err)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-400:])
    n_ok, n_err = map(int, r.stdout.split())
    assert n_ok + n_err == 400 and n_err > 50


def test_obj_polygons_are_ear_clipped_like_tinyobjloader(ref, ngp, tmp_path):
    """faces with 5 .. 12 corners: planar convex, planar concave (star-shaped), non-planar, in every axis orientation, with collinear corners -- the triangle lists
    must equal tinyobjloader's (its built-in ear clipping, restated in host/mesh_lite.hpp) bit for bit"""
    rs = np.random.default_rng(11)
    lines, n_v = [], 0
    for case in range(300):
        n = int(rs.integers(5, 13))
        ang = np.sort(rs.uniform(0, 2 * np.pi, n))
        rad = np.ones(n) if case % 3 == 0 else rs.uniform(0.3, 1.0, n)  # convex / star-shaped
        poly = np.stack([rad * np.cos(ang), rad * np.sin(ang), np.zeros(n)], 1)
        if case % 5 == 0:
            poly[:, 2] += rs.normal(0, 0.2, n)  # non-planar
        if case % 7 == 0 and n > 5:
            poly[2] = 0.5 * (poly[1] + poly[3])  # a collinear corner
        q, _ = np.linalg.qr(rs.normal(size=(3, 3)))
        rot = q if case % 2 else np.eye(3)[:, rs.permutation(3)]  # arbitrary orientation / axis-aligned planes
        poly = (poly @ rot.T + rs.normal(0, 2, 3)).astype(np.float32)
        if case % 11 == 0:
            poly = poly[::-1]  # clockwise
        lines += [f"v {p[0]:.9g} {p[1]:.9g} {p[2]:.9g}" for p in poly]
        lines.append("f " + " ".join(str(n_v + 1 + k) for k in range(n)))
        n_v += n
    f = tmp_path / "polys.obj"; f.write_text("\n".join(lines) + "\n")
    a, b = ngp.read_obj(str(f)), _ref_obj(ref, str(f))
    assert a.shape == b.shape, (a.shape, b.shape)
    bad = np.flatnonzero((a.view(np.uint32) != b.view(np.uint32)).reshape(len(a), -1).any(1))
    assert bad.size == 0, (bad.size, bad[:5].tolist())
