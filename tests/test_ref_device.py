"""CPU: the oracle's restatement of the reference's in-repository arithmetic against THE REFERENCE'S OWN DEVICE HEADERS.

oracle/_ref/libngpdev_ref.so is include/neural-graphics-primitives/{nerf_device, common_device, random_val, bounding_box}.cuh compiled for the CPU from /root/reference where
they lie (oracle/Makefile, oracle/ref_ngpdev_wrapper.cpp) against oracle/ref_shim -- a stand-in for tiny-cuda-nn's vector types and pcg32, which are absent from the mount.
Every ref_* export calls one reference function; its ora_* twin is the restatement every GPU parity test uses.  Equal bits on random inputs pin every constant, branch and
operation order the reference wrote for: stepping space, occupancy indexing and skipping, cascades, warps, activations, the seven losses, the seven lens models (rays and
the inverse projection), Sobol / Owen-scrambled sampling, box intersection, sRGB, texel reads, error-CDF sampling.  Not pinned by this: tcnn's own arithmetic (hash grid,
MLPs, optimizer: their source is absent) and tcnn's vector semantics (written the GLSL way in the shim).  Skipped when oracle/_ref is not built."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import ngp_abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libngpdev_ref.so")
F = C.c_float


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(SO):
        pytest.skip("oracle/_ref/libngpdev_ref.so not built (needs /root/reference; `make -C oracle ref`)")
    lib = C.CDLL(SO)
    for n in ("calc_dt", "advance_n_steps", "to_stepping_space", "from_stepping_space", "advance_to_next_voxel", "distance_to_next_voxel", "warp_dt", "unwarp_dt", "network_to_rgb",
              "network_to_rgb_derivative", "network_to_density", "network_to_density_derivative", "ld_random_val", "srgb_to_linear", "linear_to_srgb", "read_depth",
              "if_unoccupied_advance_to_next_occupied_voxel"):
        getattr(lib, "ref_" + n).restype = F
    lib.ref_sobol.restype = C.c_uint32
    return lib


FLOAT_FNS = ("calc_dt", "advance_n_steps", "to_stepping_space", "from_stepping_space", "advance_to_next_voxel", "distance_to_next_voxel", "warp_dt", "unwarp_dt", "network_to_rgb",
             "network_to_rgb_derivative", "network_to_density", "network_to_density_derivative", "ld_random_val", "srgb_to_linear", "linear_to_srgb", "read_depth",
             "if_unoccupied_advance_to_next_occupied_voxel")


class _ProductOnTheHost:
    """`ora_<name>` -> `ngp_host_<name>` of libngp_hip_testhooks.so (include/ngp_hip_host_hooks.h): the product's device source evaluated on the host, in the oracle's place"""

    def __init__(self, lib, ora, api):
        self._lib, self._ora, self._api = lib, ora, api
        for n in FLOAT_FNS:
            getattr(lib, "ngp_host_" + n).restype = F
        lib.ngp_host_sobol.restype = C.c_uint32; lib.ngp_host_image_idx_cdf.restype = C.c_uint32

    def __getattr__(self, name):
        lib = self._api if name in ("ora_uv_to_ray", "ora_pos_to_uv") else self._lib  # (the camera hooks are entries of the C-ABI, include/ngp_hip.h)
        if name == "ora_uv_to_ray":  # (the camera hooks predate this file and take the metadata first)
            return lambda uv, m, x, o3, d3: lib.ngp_host_uv_to_ray(m, x, uv, o3, d3)
        if name == "ora_pos_to_uv":
            return lambda p, m, x, uv2: lib.ngp_host_pos_to_uv(m, x, p, uv2)
        if name == "ora_construct_error_cdfs":  # test set-up only (a kernel, compared on the GPU and in tests/test_ref_kernels.py)
            return self._ora.ora_construct_error_cdfs
        return getattr(lib, "ngp_host_" + name[4:])


@pytest.fixture(scope="module", params=["oracle", "product_device_source_on_the_host"])
def o(request, ora):
    """the side that is held against the reference: the oracle's restatement, or the product's own device functions compiled for the host"""
    if request.param == "oracle":
        for n in FLOAT_FNS:
            getattr(ora, "ora_" + n).restype = F
        ora.ora_sobol.restype = C.c_uint32; ora.ora_image_idx_cdf.restype = C.c_uint32
        return ora
    return _ProductOnTheHost(A.load_testhooks(), ora, A.load_hip())


CONES = (0.0, 1.0 / 256.0, 1.0 / 128.0, 0.01)


def test_stepping_space(ref, o):
    """to / from_stepping_space, advance_n_steps, calc_dt (nerf_device.cuh:379-429) over the three regimes of the exponential schedule"""
    rs = np.random.default_rng(0)
    ts = np.concatenate([rs.uniform(0, 0.05, 300), rs.uniform(0, 4, 600), rs.uniform(4, 200, 300), [0.0, 1e-6, 1.0]]).astype(np.float32)
    for c in CONES:
        for t in ts:
            t = float(t)
            a, b = o.ora_to_stepping_space(F(t), F(c)), ref.ref_to_stepping_space(F(t), F(c))
            assert _bits(a) == _bits(b), ("to", t, c, a, b)
            assert _bits(o.ora_from_stepping_space(F(a), F(c))) == _bits(ref.ref_from_stepping_space(F(a), F(c))), ("from", a, c)
            assert _bits(o.ora_calc_dt(F(t), F(c))) == _bits(ref.ref_calc_dt(F(t), F(c))), ("dt", t, c)
            n = float(rs.uniform(0, 3))
            assert _bits(o.ora_advance_n_steps(F(t), F(c), F(n))) == _bits(ref.ref_advance_n_steps(F(t), F(c), F(n))), ("adv", t, c, n)
    for dt in rs.uniform(0, 0.3, 200).astype(np.float32):
        assert _bits(o.ora_warp_dt(F(dt))) == _bits(ref.ref_warp_dt(F(dt))) and _bits(o.ora_unwarp_dt(F(dt))) == _bits(ref.ref_unwarp_dt(F(dt)))


def test_occupancy_indexing_and_skipping(ref, o):
    """cascaded_grid_idx_at, mip_from_pos / mip_from_dt, density_grid_occupied_at, distance_to_next_voxel, advance_to_next_voxel, if_unoccupied_advance_to_next_occupied_voxel
    (nerf_device.cuh:317-368, 431-495): integer outputs equal, float outputs bit-equal"""
    rs = np.random.default_rng(1)
    n = 4000
    pos = np.concatenate([rs.uniform(-0.2, 1.2, (n // 2, 3)), rs.uniform(-7.5, 8.5, (n // 2, 3))]).astype(np.float32)
    pos[:8] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1.0, 0.0, 0.5], [0.99999994, 0.5, 0.5], [-1e-8, 0.5, 0.5], [0.5, 1.0000001, 0.5], [16.0, 0.5, 0.5]]
    for mip in range(8):
        a = np.zeros(n, np.uint32); b = np.zeros(n, np.uint32)
        o.ora_cascaded_grid_idx_at(_fp(pos), n, mip, a.ctypes.data_as(C.c_void_p)); ref.ref_cascaded_grid_idx_at(_fp(pos), n, mip, b.ctypes.data_as(C.c_void_p))
        assert np.array_equal(a, b), mip
    for mc in (0, 2, 7):
        a = np.zeros(n, np.uint32); b = np.zeros(n, np.uint32)
        o.ora_mip_from_pos(_fp(pos), n, mc, a.ctypes.data_as(C.c_void_p)); ref.ref_mip_from_pos(_fp(pos), n, mc, b.ctypes.data_as(C.c_void_p))
        assert np.array_equal(a, b)
        dt = rs.uniform(0, 0.25, n).astype(np.float32)
        o.ora_mip_from_dt(_fp(dt), _fp(pos), n, mc, a.ctypes.data_as(C.c_void_p)); ref.ref_mip_from_dt(_fp(dt), _fp(pos), n, mc, b.ctypes.data_as(C.c_void_p))
        # mip_from_dt ends in clamp((int)mip, exponent, (int)max_cascade) (nerf_device.cuh:459): when the step size alone asks for a cascade ABOVE max_cascade the bounds
        # cross.  tcnn's scalar clamp tests the lower bound first (a < b ? b : (c < a ? c : a)) -> `exponent`, a pooled level above max_cascade; that is the shim's default, the
        # oracle and the HIP kernels since round 4 (and what the pre-tcnn code computed: min(NERF_CASCADES() - 1, max(exponent, mip))).
        # tests/test_ref_kernels.py::test_k1_scalar_clamp_with_crossed_bounds measures what GLSL's min(max()) would change.
        exponent = np.frexp(dt * np.float32(256.0))[1]
        degenerate = (dt * np.float32(256.0) >= 1.0) & (exponent > mc)
        assert np.array_equal(a, b) and (degenerate.sum() > 100 or mc == 7) and np.all(a[degenerate] == exponent[degenerate]) and a.max() <= 7
    bf = rs.integers(0, 256, 128 ** 3 // 8 * 8, dtype=np.uint8)
    bf[rs.uniform(size=bf.size) < 0.6] = 0
    bfp = bf.ctypes.data_as(C.c_void_p)
    for i in range(1500):
        p = pos[i]; d = rs.normal(size=3).astype(np.float32); d /= np.linalg.norm(d)
        mip = int(rs.integers(0, 8))
        assert o.ora_density_grid_occupied_at(_fp(p), bfp, mip) == ref.ref_density_grid_occupied_at(_fp(p), bfp, mip)
        res = float(128 >> int(rs.integers(0, 4)))
        assert _bits(o.ora_distance_to_next_voxel(_fp(p), _fp(d), F(res))) == _bits(ref.ref_distance_to_next_voxel(_fp(p), _fp(d), F(res)))
        t, c = float(rs.uniform(0, 3)), CONES[i % 4]
        assert _bits(o.ora_advance_to_next_voxel(F(t), F(c), _fp(p), _fp(d), mip)) == _bits(ref.ref_advance_to_next_voxel(F(t), F(c), _fp(p), _fp(d), mip)), (i, t, c)
    # axis-parallel rays: a direction component that is exactly 0 goes through sign(0), which is +1 in tcnn (copysign) -- the voxel step stays finite and positive
    for i in range(8, 608):  # (the first eight positions sit on voxel faces, where a zero distance is the right answer)
        p = pos[i]; d = rs.normal(size=3).astype(np.float32); d[i % 3] = 0.0 if i % 2 else -0.0; d /= np.linalg.norm(d)
        a, b = o.ora_distance_to_next_voxel(_fp(p), _fp(d), F(128.0)), ref.ref_distance_to_next_voxel(_fp(p), _fp(d), F(128.0))
        assert _bits(a) == _bits(b) and a > 0 and np.isfinite(a), (i, a, b)
        assert _bits(o.ora_advance_to_next_voxel(F(0.7), F(CONES[i % 4]), _fp(p), _fp(d), 0)) == _bits(ref.ref_advance_to_next_voxel(F(0.7), F(CONES[i % 4]), _fp(p), _fp(d), 0))
    for scale, max_mip in ((1, 0), (4, 2), (16, 4)):
        box = A.scene_aabb(scale)
        for i in range(400):
            org = rs.uniform(-0.5, 1.5, 3).astype(np.float32); d = (np.array([0.5, 0.5, 0.5], np.float32) + rs.normal(0, 0.4, 3).astype(np.float32) - org); d = (d / np.linalg.norm(d)).astype(np.float32)
            t, c = float(rs.uniform(0, 1.5)), CONES[i % 4]
            a = o.ora_if_unoccupied_advance_to_next_occupied_voxel(F(t), F(c), _fp(org), _fp(d), bfp, 0, max_mip, C.byref(box))
            b = ref.ref_if_unoccupied_advance_to_next_occupied_voxel(F(t), F(c), _fp(org), _fp(d), bfp, 0, max_mip, C.byref(box))
            assert _bits(a) == _bits(b), (scale, i, a, b)


def test_warps_activations_losses(ref, o):
    """warp_position / unwarp / warp_direction, network_to_rgb / density and their derivatives for every activation, loss_and_gradient for the seven losses
    (nerf_device.cuh:75-143, 204-310, 601-616)"""
    rs = np.random.default_rng(2)
    for scale in (1, 4, 16):
        box = A.scene_aabb(scale)
        for _ in range(200):
            p = rs.uniform(-8, 9, 3).astype(np.float32); a = np.zeros(3, np.float32); b = np.zeros(3, np.float32)
            o.ora_warp_position(_fp(p), C.byref(box), _fp(a)); ref.ref_warp_position(_fp(p), C.byref(box), _fp(b)); assert np.array_equal(_bits(a), _bits(b))
            o.ora_unwarp_position(_fp(p), C.byref(box), _fp(a)); ref.ref_unwarp_position(_fp(p), C.byref(box), _fp(b)); assert np.array_equal(_bits(a), _bits(b))
            o.ora_warp_direction(_fp(p), _fp(a)); ref.ref_warp_direction(_fp(p), _fp(b)); assert np.array_equal(_bits(a), _bits(b))
            assert o.ora_aabb_contains(C.byref(box), _fp(p)) == ref.ref_aabb_contains(C.byref(box), _fp(p))
    vals = np.concatenate([rs.normal(0, 4, 400), [0.0, -0.0, 15.0, -15.0, 20.0, -20.0, 88.0, -100.0]]).astype(np.float32)
    for act in range(4):
        for v in vals:
            for fn in ("network_to_rgb", "network_to_rgb_derivative", "network_to_density", "network_to_density_derivative"):
                a, b = getattr(o, "ora_" + fn)(F(float(v)), act), getattr(ref, "ref_" + fn)(F(float(v)), act)
                assert _bits(a) == _bits(b), (fn, act, float(v), a, b)
    for loss in range(7):
        for _ in range(300):
            tg = rs.uniform(-0.2, 1.5, 3).astype(np.float32); pr = rs.uniform(-0.2, 1.5, 3).astype(np.float32)
            if _ % 17 == 0:
                pr = tg.copy()
            la, ga, lb, gb = (np.zeros(3, np.float32) for _ in range(4))
            o.ora_loss_and_gradient(_fp(tg), _fp(pr), loss, _fp(la), _fp(ga)); ref.ref_loss_and_gradient(_fp(tg), _fp(pr), loss, _fp(lb), _fp(gb))
            assert np.array_equal(_bits(la), _bits(lb)) and np.array_equal(_bits(ga), _bits(gb)), (loss, tg.tolist(), pr.tolist(), la.tolist(), lb.tolist(), ga.tolist(), gb.tolist())


def _meta(rs, lens, res=(80, 60)):
    m = A.ImageMeta()
    m.resolution[0], m.resolution[1] = res
    m.focal_length[0] = float(rs.uniform(40, 120)); m.focal_length[1] = float(rs.uniform(40, 120))
    m.principal_point[0] = float(rs.uniform(0.4, 0.6)); m.principal_point[1] = float(rs.uniform(0.4, 0.6))
    m.lens_mode = lens
    p = [0.0] * 7
    if lens == A.LENS_OPENCV:
        p[:4] = [float(rs.normal(0, 0.05)), float(rs.normal(0, 0.02)), float(rs.normal(0, 0.002)), float(rs.normal(0, 0.002))]
    elif lens == A.LENS_OPENCV_FISHEYE:
        p[:4] = [float(rs.normal(0, 0.03)), float(rs.normal(0, 0.01)), float(rs.normal(0, 0.003)), float(rs.normal(0, 0.001))]
    elif lens == A.LENS_FTHETA:
        p[:7] = [0.0, float(rs.uniform(0.9, 1.1)) / m.focal_length[0], float(rs.normal(0, 1e-6)), float(rs.normal(0, 1e-8)), 0.0, float(res[0]), float(res[1])]
    for k in range(7):
        m.lens_params[k] = p[k]
    return m


@pytest.mark.parametrize("lens", list(range(7)))
def test_lens_models(ref, o, lens):
    """uv_to_ray as generate_training_samples_nerf calls it and pos_to_uv (common_device.cuh:268-577) for all seven lens models, Newton-iterated undistortion included:
    origins, directions, validity and the inverse projection, bit for bit -- uv outside the unit square included (the default Foveation both functions apply clamps it:
    this pin is what found the clamp missing from the restatement).  One exception, stated: the Newton step of the two iterated lenses multiplies by tcnn's inverse(mat2),
    whose rounding is tcnn's (absent); the oracle and the HIP header solve the same 2 x 2 system by Cramer's rule with a division.  The central-difference iteration amplifies
    that last-bit difference, so for OpenCV / OpenCVFisheye the directions must agree to 1e-5 relative and in the large majority of rays to the bit."""
    rs = np.random.default_rng(10 + lens)
    newton = lens in (A.LENS_OPENCV, A.LENS_OPENCV_FISHEYE)
    n_rays = n_exact = 0
    for trial in range(12):
        m = _meta(rs, lens)
        q, _ = np.linalg.qr(rs.normal(size=(3, 3)))
        x = np.concatenate([q.T.reshape(-1), rs.normal(0, 1, 3)]).astype(np.float32)  # columns of the rotation, then the position
        for _ in range(60):
            uv = rs.uniform(-0.05, 1.05, 2).astype(np.float32)
            oa, da, ob, db = (np.zeros(3, np.float32) for _ in range(4))
            va = o.ora_uv_to_ray(_fp(uv), C.byref(m), _fp(x), _fp(oa), _fp(da)); vb = ref.ref_uv_to_ray(_fp(uv), C.byref(m), _fp(x), _fp(ob), _fp(db))
            assert va == vb, (lens, uv.tolist())
            if va:
                assert np.array_equal(_bits(oa), _bits(ob)), (lens, uv.tolist(), oa.tolist(), ob.tolist())
                exact = np.array_equal(_bits(da), _bits(db))
                n_rays += 1; n_exact += int(exact)
                if newton:
                    assert np.linalg.norm(da.astype(np.float64) - db) <= 1e-5 * np.linalg.norm(db), (lens, uv.tolist(), da.tolist(), db.tolist())
                else:
                    assert exact, (lens, uv.tolist(), da.tolist(), db.tolist())
            if lens != A.LENS_FTHETA:  # (f-theta has no forward mapping: the reference asserts)
                p = (x[9:12] + q.T[2] * rs.uniform(0.5, 3) + rs.normal(0, 0.5, 3)).astype(np.float32)
                ua, ub = np.zeros(2, np.float32), np.zeros(2, np.float32)
                o.ora_pos_to_uv(_fp(p), C.byref(m), _fp(x), _fp(ua)); ref.ref_pos_to_uv(_fp(p), C.byref(m), _fp(x), _fp(ub))
                assert np.array_equal(_bits(ua), _bits(ub)), (lens, p.tolist(), ua.tolist(), ub.tolist())
    assert n_rays > 500 and n_exact >= (0.9 if newton else 1.0) * n_rays, (n_exact, n_rays)


def test_sampling_sequences_colour_and_texels(ref, o):
    """sobol / ld_random_val / ld_random_pixel_offset (random_val.cuh:162-325), sRGB conversions, box / ray intersection, byte-image and depth texel reads, image_idx and
    sample_cdf_2d with CDFs (nerf_device.cuh:497-599)"""
    rs = np.random.default_rng(3)
    for i in list(range(300)) + rs.integers(0, 2 ** 32, 300).tolist():
        for dim in (0, 1):  # the two dimensions the path draws (ld_random_val's default dimension 0 and ld_random_pixel_offset's pair); the oracle holds no tables for 2..4
            assert o.ora_sobol(int(i), dim) == ref.ref_sobol(int(i), dim)
        seed = int(rs.integers(0, 2 ** 32))
        assert _bits(o.ora_ld_random_val(int(i), seed, int(i) % 2)) == _bits(ref.ref_ld_random_val(int(i), seed, int(i) % 2))
    for spp in range(64):
        a, b = np.zeros(2, np.float32), np.zeros(2, np.float32)
        o.ora_ld_random_pixel_offset(spp, _fp(a)); ref.ref_ld_random_pixel_offset(spp, _fp(b)); assert np.array_equal(_bits(a), _bits(b))
    for v in np.concatenate([rs.uniform(-0.1, 1.2, 500), [0.0, 0.04045, 0.0031308, 1.0]]).astype(np.float32):
        assert _bits(o.ora_srgb_to_linear(F(float(v)))) == _bits(ref.ref_srgb_to_linear(F(float(v)))) and _bits(o.ora_linear_to_srgb(F(float(v)))) == _bits(ref.ref_linear_to_srgb(F(float(v))))
    for scale in (1, 4):
        box = A.scene_aabb(scale)
        for _ in range(400):
            org = rs.uniform(-3, 4, 3).astype(np.float32); d = rs.normal(size=3).astype(np.float32); d /= np.linalg.norm(d)
            if _ % 9 == 0:
                d[int(rs.integers(0, 3))] = 0.0
            a, b = np.zeros(2, np.float32), np.zeros(2, np.float32)
            o.ora_aabb_ray_intersect(C.byref(box), _fp(org), _fp(d), _fp(a)); ref.ref_aabb_ray_intersect(C.byref(box), _fp(org), _fp(d), _fp(b))
            assert np.array_equal(_bits(a), _bits(b)), (org.tolist(), d.tolist(), a.tolist(), b.tolist())
    w, h = 23, 17
    img = rs.integers(0, 256, (h, w, 4), dtype=np.uint8); img[0, 0] = [0, 0, 0, 0]; img[1, 1, 3] = 0  # incl. the "masked away" marker pixels
    dep = rs.uniform(0, 5, (h, w)).astype(np.float32)
    res = (C.c_int32 * 2)(w, h)
    for _ in range(500):
        uv = rs.uniform(-0.1, 1.1, 2).astype(np.float32)
        a, b = np.zeros(4, np.float32), np.zeros(4, np.float32)
        o.ora_read_rgba_byte(_fp(uv), res, img.ctypes.data_as(C.c_void_p), _fp(a)); ref.ref_read_rgba_byte(_fp(uv), res, img.ctypes.data_as(C.c_void_p), _fp(b))
        assert np.array_equal(_bits(a), _bits(b)), (uv.tolist(), a.tolist(), b.tolist())
        assert _bits(o.ora_read_depth(_fp(uv), res, _fp(dep))) == _bits(ref.ref_read_depth(_fp(uv), res, _fp(dep)))
    # error CDFs: the oracle's own construction, then image / pixel selection through the reference's functions
    n_img, ch, cw = 7, 12, 15
    err = rs.uniform(0, 1e-3, (n_img, ch, cw)).astype(np.float32); err[2, 3:5, 6:9] += 0.05
    cxy = np.zeros_like(err); cy = np.zeros((n_img, ch), np.float32); ci = np.zeros(n_img, np.float32)
    o.ora_construct_error_cdfs(n_img, cw, ch, _fp(err), _fp(cxy), _fp(cy), _fp(ci))
    pa, pb = C.c_float(), C.c_float()
    ref.ref_image_idx.restype = C.c_uint32
    for i in rs.integers(0, 2 ** 20, 600).tolist():
        assert o.ora_image_idx_cdf(int(i), n_img, _fp(ci), C.byref(pa)) == ref.ref_image_idx(int(i), 4096, 0, n_img, _fp(ci), C.byref(pb)) and _bits(pa.value) == _bits(pb.value)
    cres = (C.c_int32 * 2)(cw, ch)
    for _ in range(600):
        smp = rs.uniform(0, 1, 2).astype(np.float32); img_i = int(rs.integers(0, n_img))
        ua, ub = np.zeros(2, np.float32), np.zeros(2, np.float32); pa, pb = C.c_float(1.0), C.c_float(1.0)
        o.ora_sample_cdf_2d(_fp(smp), img_i, cres, _fp(cxy), _fp(cy), _fp(ua), C.byref(pa)); ref.ref_sample_cdf_2d(_fp(smp), img_i, cres, _fp(cxy), _fp(cy), _fp(ub), C.byref(pb))
        assert np.array_equal(_bits(ua), _bits(ub)) and _bits(pa.value) == _bits(pb.value)
