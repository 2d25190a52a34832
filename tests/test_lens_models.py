"""CPU: the seven lens models of the reference's camera (common_device.cuh:268-577).

Three implementations are compared without a GPU: the device header's uv_to_ray / pos_to_uv compiled for the HOST (the same source the
kernels compile, through the test hooks ngp_host_uv_to_ray / ngp_host_pos_to_uv of the C-ABI), the oracle's restatement, and an independent
numpy float64 model of every lens written from the formulas.  Plus the defining property: pos_to_uv(origin + t * direction) == uv wherever
the lens has a forward mapping."""
import ctypes as C
import math

import numpy as np
import pytest

import ngp_abi as A

LENSES = {"perspective": A.LENS_PERSPECTIVE, "opencv": A.LENS_OPENCV, "opencv_fisheye": A.LENS_OPENCV_FISHEYE, "ftheta": A.LENS_FTHETA,
          "latlong": A.LENS_LATLONG, "equirectangular": A.LENS_EQUIRECTANGULAR, "orthographic": A.LENS_ORTHOGRAPHIC}
W, H = 1280, 720


def _meta(mode):
    m = A.ImageMeta()
    m.lens_mode = mode
    m.resolution[0], m.resolution[1] = W, H
    m.focal_length[0], m.focal_length[1] = 900.0, 880.0
    m.principal_point[0], m.principal_point[1] = 0.52, 0.47
    params = {A.LENS_OPENCV: [0.11, -0.05, 0.002, -0.003], A.LENS_OPENCV_FISHEYE: [0.05, -0.02, 0.004, -0.001],
              A.LENS_FTHETA: [0.0, 1.1e-3, 2.0e-8, -1.0e-11, 0.0, float(W), float(H)]}.get(mode, [])
    for k, v in enumerate(params):
        m.lens_params[k] = v
    return m


def _xform():
    # an arbitrary rigid camera: rotation (columns) + position, column-major 4x3 like ngp_xform.start
    a, b, c = 0.3, -0.7, 1.1
    rx = np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])
    ry = np.array([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]])
    rz = np.array([[math.cos(c), -math.sin(c), 0], [math.sin(c), math.cos(c), 0], [0, 0, 1]])
    r = rz @ ry @ rx
    x = np.concatenate([r[:, 0], r[:, 1], r[:, 2], [0.4, 0.55, 0.6]]).astype(np.float32)
    return (C.c_float * 12)(*x.tolist()), r, np.array([0.4, 0.55, 0.6])


def _uvs(n=400):
    rng = np.random.default_rng(5)
    return rng.uniform(0.02, 0.98, (n, 2)).astype(np.float32)


def _model(mode, m, uv):
    """independent float64 model: camera-space direction and head position for one uv (None = no ray)"""
    cx, cy = m.principal_point[0], m.principal_point[1]
    fx, fy = m.focal_length[0], m.focal_length[1]
    u, v = float(uv[0]), float(uv[1])
    px, py = (u - cx) * W / fx, (v - cy) * H / fy
    head = np.zeros(3)
    if mode == A.LENS_FTHETA:
        p = [m.lens_params[k] for k in range(7)]
        xp, yp = (u - cx) * p[5], (v - cy) * p[6]
        n = math.hypot(xp, yp)
        alpha = p[0] + n * (p[1] + n * (p[2] + n * (p[3] + n * p[4])))
        if math.cos(alpha) <= 0 or n == 0:
            return None, head
        return np.array([math.sin(alpha) * xp / n, math.sin(alpha) * yp / n, math.cos(alpha)]), head
    if mode == A.LENS_LATLONG:
        th, ph = (v - 0.5) * math.pi, (u - 0.5) * 2 * math.pi
        return np.array([math.sin(ph) * math.cos(th), math.sin(th), math.cos(ph) * math.cos(th)]), head
    if mode == A.LENS_EQUIRECTANGULAR:
        ct, ph = (v - 0.5) * 2, (u - 0.5) * 2 * math.pi
        st = math.sqrt(max(1 - ct * ct, 0))
        return np.array([math.sin(ph) * st, ct, math.cos(ph) * st]), head
    if mode == A.LENS_ORTHOGRAPHIC:
        return np.array([0.0, 0.0, 1.0]), np.array([px, py, 0.0])
    if mode == A.LENS_PERSPECTIVE:
        return np.array([px, py, 1.0]), head

    def dist(x, y):  # distorted image point of the undistorted (x, y)
        if mode == A.LENS_OPENCV:
            k1, k2, p1, p2 = [m.lens_params[k] for k in range(4)]
            r2 = x * x + y * y
            rad = k1 * r2 + k2 * r2 * r2
            return x + x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x), y + y * rad + 2 * p2 * x * y + p1 * (r2 + 2 * y * y)
        k = [m.lens_params[i] for i in range(4)]
        r = math.hypot(x, y)
        if r < 1e-15:
            return x, y
        th = math.atan(r)
        thd = th * (1 + k[0] * th ** 2 + k[1] * th ** 4 + k[2] * th ** 6 + k[3] * th ** 8)
        return x * thd / r, y * thd / r
    x, y = px, py  # fixed-point iteration with an analytic-free Newton (secant Jacobian) in float64
    for _ in range(200):
        fx0, fy0 = dist(x, y)
        e = 1e-7
        j00 = (dist(x + e, y)[0] - dist(x - e, y)[0]) / (2 * e); j01 = (dist(x, y + e)[0] - dist(x, y - e)[0]) / (2 * e)
        j10 = (dist(x + e, y)[1] - dist(x - e, y)[1]) / (2 * e); j11 = (dist(x, y + e)[1] - dist(x, y - e)[1]) / (2 * e)
        det = j00 * j11 - j01 * j10
        rx_, ry_ = fx0 - px, fy0 - py
        dx, dy = (j11 * rx_ - j01 * ry_) / det, (-j10 * rx_ + j00 * ry_) / det
        x, y = x - dx, y - dy
        if dx * dx + dy * dy < 1e-28:
            break
    return np.array([x, y, 1.0]), head


@pytest.mark.parametrize("name", sorted(LENSES))
def test_uv_to_ray_three_ways(ora, name):
    mode = LENSES[name]
    hip = A.load_hip()  # host-side hooks: no GPU needed
    m = _meta(mode)
    x12, r, cam_pos = _xform()
    n_valid = 0
    for uv in _uvs():
        uvc = (C.c_float * 2)(float(uv[0]), float(uv[1]))
        oh, dh, oo, do = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
        ok_h = hip.ngp_host_uv_to_ray(C.byref(m), x12, uvc, oh, dh)
        ok_o = ora.ora_uv_to_ray(uvc, C.byref(m), x12, oo, do)
        assert ok_h == ok_o
        dir_m, head = _model(mode, m, uv)
        assert (dir_m is not None) == bool(ok_h)
        # device header on the host vs the oracle: the same float32 formulas (libm calls may differ in the last ulps)
        assert np.allclose(np.array(oh[:]), np.array(oo[:]), rtol=0, atol=2e-6) and np.allclose(np.array(dh[:]), np.array(do[:]), rtol=2e-6, atol=2e-6)
        if not ok_h:
            continue
        n_valid += 1
        # vs the float64 model (direction up to the float32 rounding of the pipeline and the Newton tolerance of the undistortion)
        d_m = r @ dir_m
        o_m = r @ head + cam_pos
        assert np.allclose(np.array(dh[:]), d_m, rtol=2e-5, atol=2e-5), (name, uv, dh[:], d_m)
        assert np.allclose(np.array(oh[:]), o_m, rtol=0, atol=2e-6)
    assert n_valid > 50  # (f-theta: the uvs inside its field of view)


@pytest.mark.parametrize("name", sorted(set(LENSES) - {"ftheta"}))  # f-theta has no forward mapping (the reference asserts)
def test_pos_to_uv_inverts_uv_to_ray(ora, name):
    mode = LENSES[name]
    hip = A.load_hip()
    m = _meta(mode)
    x12, _, _ = _xform()
    worst = 0.0
    for i, uv in enumerate(_uvs(200)):
        uvc = (C.c_float * 2)(float(uv[0]), float(uv[1]))
        o, d = (C.c_float * 3)(), (C.c_float * 3)()
        assert hip.ngp_host_uv_to_ray(C.byref(m), x12, uvc, o, d) == 1
        t = 0.3 + 0.01 * i
        pos = (C.c_float * 3)(*[o[k] + t * d[k] for k in range(3)])
        back_h, back_o = (C.c_float * 2)(), (C.c_float * 2)()
        hip.ngp_host_pos_to_uv(C.byref(m), x12, pos, back_h)
        ora.ora_pos_to_uv(pos, C.byref(m), x12, back_o)
        assert abs(back_h[0] - back_o[0]) <= 2e-6 and abs(back_h[1] - back_o[1]) <= 2e-6
        worst = max(worst, abs(back_h[0] - uv[0]), abs(back_h[1] - uv[1]))
    assert worst <= 2e-5, (name, worst)


def test_rolling_shutter_camera_interpolation(ora):
    """get_xform_given_rolling_shutter / camera_slerp (common_device.cuh:665-674; slerp(mat3, mat3, t) from tiny-cuda-nn's published vec.h): the device
    header compiled for the host vs the oracle's restatement vs scipy's quaternion slerp in float64.  t = a + b u + c v + d motionblur_time; the rotation
    follows the short great arc, the position is interpolated linearly; a frame without motion data keeps its matrix bit for bit."""
    from scipy.spatial.transform import Rotation, Slerp
    lib = A.load_hip()
    rng = np.random.default_rng(11)
    worst_dev_ora = worst_model = 0.0
    for trial in range(40):
        r0 = Rotation.from_rotvec(rng.normal(size=3) * 1.5)
        r1 = r0 * Rotation.from_rotvec(rng.normal(size=3) * (0.02 if trial % 2 else 1.0))  # small and large camera motion
        p0, p1 = rng.uniform(-1, 2, 3), rng.uniform(-1, 2, 3)
        X = A.Xform()
        for c in range(3):
            for r in range(3):
                X.start[c * 3 + r] = float(r0.as_matrix()[r, c]); X.end[c * 3 + r] = float(r1.as_matrix()[r, c])
        for r in range(3):
            X.start[9 + r] = float(p0[r]); X.end[9 + r] = float(p1[r])
        rs = (C.c_float * 4)(*rng.uniform(-0.2, 0.5, 4).astype(np.float32))
        uv = (C.c_float * 2)(*rng.uniform(0, 1, 2).astype(np.float32))
        mb = float(np.float32(rng.uniform(0, 1)))
        t = float(np.float32(rs[0]) + np.float32(rs[1]) * np.float32(uv[0]) + np.float32(rs[2]) * np.float32(uv[1]) + np.float32(rs[3]) * np.float32(mb))
        dev = (C.c_float * 12)(); orc = (C.c_float * 12)()
        assert lib.ngp_host_xform_given_rolling_shutter(C.byref(X), rs, uv, C.c_float(mb), dev) == 0
        ora.ora_xform_given_rolling_shutter(C.byref(X), rs, uv, C.c_float(mb), orc)
        d, o = np.array(dev[:]), np.array(orc[:])
        worst_dev_ora = max(worst_dev_ora, float(np.abs(d - o).max()))
        # float64 model: scipy's slerp extrapolates outside [0, 1] the same way (angle * t along the same arc)
        rot = (r0 * Rotation.from_rotvec((r0.inv() * r1).as_rotvec() * t)).as_matrix()
        pos = p0 * (1 - t) + p1 * t
        model = np.concatenate([rot[:, 0], rot[:, 1], rot[:, 2], pos])
        worst_model = max(worst_model, float(np.abs(d - model).max()))
        R = d[:9].reshape(3, 3).T
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-5
    assert worst_dev_ora <= 1e-6, worst_dev_ora   # same arithmetic, two implementations (acosf / sinf of the same libm here)
    assert worst_model <= 5e-6, worst_model       # fp32 vs the float64 model
    # no motion data: the matrix comes back untouched (csrc/ngp_device.hpp explains the deviation from the reference's quaternion round trip)
    X = A.Xform()
    vals = rng.normal(size=12).astype(np.float32)
    for k in range(12):
        X.start[k] = X.end[k] = float(vals[k])
    out = (C.c_float * 12)()
    lib.ngp_host_xform_given_rolling_shutter(C.byref(X), (C.c_float * 4)(0, 0, 0, 0), (C.c_float * 2)(0.3, 0.7), C.c_float(0.5), out)
    assert np.array_equal(np.array(out[:], np.float32).view(np.uint32), vals.view(np.uint32))
