"""SDF primitive (BASELINE config 4): ground-truth distances through the triangle BVH, training-sample generation, trainer, IoU.
CPU: the brute-force oracle (oracle/ora_sdf.hpp) against an analytic shape.  GPU: BVH distances == brute force bit for bit, sample
positions == oracle, training on armadillo.obj (the reference's mesh, staged under _ref_data/; an icosphere otherwise)."""
import ctypes as C
import os

import numpy as np
import pytest

import ngp_abi as A
from common import ptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def icosphere(subdiv=3, radius=0.3, center=(0.5, 0.5, 0.5)):
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9),
         (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(x, np.float64) / np.linalg.norm(x) for x in v]
    for _ in range(subdiv):
        nf, cache = [], {}
        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]; v.append(m / np.linalg.norm(m)); cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    V = np.array(v) * radius + np.array(center)
    return np.ascontiguousarray(V[np.array(f)].astype(np.float32))  # [n_tri, 3, 3], outward orientation


def test_brute_force_oracle_on_a_sphere(ora):
    """signed distance to a finely tessellated sphere = |p - c| - r up to the tessellation error; sign from the 32 stab rays"""
    tris = icosphere(3)  # 1280 triangles
    rng = np.random.default_rng(0)
    p = rng.uniform(0.05, 0.95, (4000, 3)).astype(np.float32)
    out = np.zeros(len(p), np.float32)
    ora.ora_sdf_signed_distance(ptr(tris), len(tris), ptr(p), len(p), None, ptr(out))
    ref = np.linalg.norm(p - 0.5, axis=1) - 0.3
    assert np.abs(out - ref).max() < 4e-3  # chord sag of the level-3 icosphere: r (1 - cos(pi / 36)) ~ 1.1e-3 (+ facet corners)
    clear = np.abs(ref) > 5e-3
    assert np.array_equal(np.sign(out[clear]), np.sign(ref[clear]))
    assert (out < 0).sum() > 200 and (out > 0).sum() > 2000


def _mesh():
    obj = os.path.join(ROOT, "_ref_data", "data", "sdf", "armadillo.obj")
    if os.path.exists(obj):
        import pyngp
        return np.ascontiguousarray(pyngp.read_obj(obj)), "armadillo.obj"
    return icosphere(4, 0.4, (0.1, -0.2, 0.3)), "icosphere"


def _dev_read(ptr_value, n, dtype):
    out = np.empty(n, dtype)
    rt = C.CDLL("libamdhip64.so")
    assert rt.hipMemcpy(ptr(out), C.c_void_p(ptr_value), C.c_size_t(out.nbytes), 2) == 0
    return out


@pytest.mark.gpu
def test_sdf_ground_truth_and_samples_match_brute_force(ora, hip):
    import torch
    tris, name = _mesh()
    verts = tris.reshape(-1, 3).copy()
    box = A.Aabb(); scale = C.c_float()
    A.check(hip, hip.ngp_sdf_normalize_mesh_host(ptr(verts), C.c_uint64(len(verts)), C.byref(box), C.byref(scale)))
    assert 0 <= verts.min() and verts.max() <= 1 and abs((verts.max(0) - verts.min(0)).max() - (1 - 2 * 0.005 * np.linalg.norm((tris.reshape(-1, 3).max(0) - tris.reshape(-1, 3).min(0))) / scale.value)) < 1e-3
    tn = np.ascontiguousarray(verts.reshape(-1, 3, 3))
    cfg = A.sdf_encmlp_config()
    hh = C.c_void_p(); A.check(hip, hip.ngp_encmlp_create(C.byref(cfg), C.c_uint64(1337), C.byref(hh)))
    o = A.default_sdf_options(batch_size=1 << 13)
    t = C.c_void_p(); A.check(hip, hip.ngp_sdf_create(hh, ptr(tn), len(tn), box, C.byref(o), C.byref(t)))
    # (a) BVH signed distance of arbitrary points == brute force over all triangles, bit for bit (the BVH only prunes)
    rng = np.random.default_rng(3)
    n = 3000
    p = rng.uniform(0.0, 1.0, (n, 3)).astype(np.float32)
    p[: n // 2] = (tn.reshape(-1, 3)[rng.integers(0, len(tn) * 3, n // 2)] + rng.normal(0, 0.01, (n // 2, 3))).astype(np.float32)  # near the surface
    pd = torch.from_numpy(p).cuda(); od = torch.zeros(n, dtype=torch.float32, device="cuda")
    A.check(hip, hip.ngp_sdf_signed_distance(t, None, C.c_void_p(pd.data_ptr()), n, C.c_void_p(od.data_ptr())))
    torch.cuda.synchronize()
    ref = np.zeros(n, np.float32)
    ora.ora_sdf_signed_distance(ptr(tn), len(tn), ptr(p), n, None, ptr(ref))
    got = od.cpu().numpy()
    same = got.view(np.uint32) == ref.view(np.uint32)
    print(f"{name}: {len(tn)} triangles, {same.sum()} / {n} signed distances bit-identical, max |delta| {np.abs(got - ref).max():.2e}, inside {int((ref < 0).sum())}")
    assert np.abs(np.abs(got) - np.abs(ref)).max() <= 1e-7 and (np.sign(got) == np.sign(ref)).mean() >= 0.999  # sincosf of the stab directions: device vs glibc
    # (b) one training batch: positions / upper bounds == the oracle's on the device's (BVH-ordered) triangles and cdf; distances == brute force
    A.check(hip, hip.ngp_sdf_train(t, None, 1)); torch.cuda.synchronize()
    pp, dp = C.c_void_p(), C.c_void_p(); hip.ngp_sdf_batch_ptrs(t, C.byref(pp), C.byref(dp))
    B = o.batch_size
    pos, dist = _dev_read(pp.value, B * 3, np.float32).reshape(B, 3), _dev_read(dp.value, B, np.float32)
    n_exact, n_surface = B // 8 * 4, B // 8 * 7
    assert np.all(dist[:n_exact] == 0)
    exact_d = np.zeros(256, np.float32)
    ora.ora_sdf_signed_distance(ptr(tn), len(tn), ptr(np.ascontiguousarray(pos[:256])), 256, None, ptr(exact_d))
    assert np.abs(exact_d).max() < 2e-6  # samples "on the surface" are on the surface
    lo, hi = np.array(box.min) - 1e-6, np.array(box.max) + 1e-6
    assert np.all(pos[n_surface:] >= lo) and np.all(pos[n_surface:] <= hi)
    off = np.linalg.norm(pos[n_exact:n_surface] - pos[n_exact:n_surface].mean(0), axis=1)
    sub = slice(n_exact, n_exact + 1500)
    bd = np.zeros(1500, np.float32)
    ora.ora_sdf_signed_distance(ptr(tn), len(tn), ptr(np.ascontiguousarray(pos[sub])), 1500, None, ptr(bd))
    assert np.abs(np.abs(dist[sub]) - np.abs(bd)).max() <= 1e-7 and (np.sign(dist[sub]) == np.sign(bd)).mean() >= 0.995
    assert np.abs(dist[sub]).max() < 0.02 and np.median(np.abs(dist[sub])) < 2e-3  # logistic offsets of sigma = 0.866 / 1024
    sub2 = slice(n_surface, n_surface + 500)
    bu = np.zeros(500, np.float32)
    ora.ora_sdf_signed_distance(ptr(tn), len(tn), ptr(np.ascontiguousarray(pos[sub2])), 500, None, ptr(bu))
    assert np.abs(np.abs(dist[sub2]) - np.abs(bu)).max() <= 1e-7 and (np.sign(dist[sub2]) == np.sign(bu)).mean() >= 0.995
    hip.ngp_sdf_destroy(t); hip.ngp_encmlp_destroy(hh)


@pytest.mark.gpu
def test_sdf_trainer_learns_the_mesh(hip):
    """train_sdf end to end (BASELINE config 4: L = 16, F = 2, T = 2^19, MLP 2x64, MAPE, batch 2^18): IoU of the learned sign vs the mesh"""
    import time
    import torch
    tris, name = _mesh()
    verts = tris.reshape(-1, 3).copy()
    box = A.Aabb(); scale = C.c_float()
    A.check(hip, hip.ngp_sdf_normalize_mesh_host(ptr(verts), C.c_uint64(len(verts)), C.byref(box), C.byref(scale)))
    tn = np.ascontiguousarray(verts.reshape(-1, 3, 3))
    cfg = A.sdf_encmlp_config()
    hh = C.c_void_p(); A.check(hip, hip.ngp_encmlp_create(C.byref(cfg), C.c_uint64(1337), C.byref(hh)))
    o = A.default_sdf_options()
    t = C.c_void_p(); A.check(hip, hip.ngp_sdf_create(hh, ptr(tn), len(tn), box, C.byref(o), C.byref(t)))
    iou0 = C.c_double(); A.check(hip, hip.ngp_sdf_iou(t, 1 << 18, C.byref(iou0)))
    A.check(hip, hip.ngp_sdf_train(t, None, 10)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    A.check(hip, hip.ngp_sdf_train(t, None, 490)); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    iou, loss = C.c_double(), C.c_float()
    A.check(hip, hip.ngp_sdf_iou(t, 1 << 20, C.byref(iou))); A.check(hip, hip.ngp_sdf_loss(t, None, C.byref(loss)))
    print(f"{name} ({len(tn)} triangles): IoU {iou0.value:.3f} -> {iou.value:.4f} after 500 steps, MAPE {loss.value:.4f}, {490 * o.batch_size / dt / 1e6:.1f} M samples/s ({dt / 490 * 1e3:.2f} ms/step incl. sample generation)")
    assert np.isfinite(loss.value) and iou.value > 0.95
    hip.ngp_sdf_destroy(t); hip.ngp_encmlp_destroy(hh)


@pytest.mark.gpu
def test_sdf_batches_ahead_equal_the_serial_loop(hip):
    """ngp_sdf_train generates the batches a group at a time on a side stream, ahead of the training steps and across calls (one ground-truth launch per group); the batches, their
    order and the rng positions other consumers see (calculate_iou) must be the serial loop's (generate -> train, testbed_sdf.cu:1580-1635): last trained batch bit-identical after
    every call, for group sizes 1 / 3 / 4 / 16, with one-step calls, calls longer than a group and an IoU evaluation (which draws from the trainer's rng) in between."""
    import torch
    tris, name = _mesh()
    verts = tris.reshape(-1, 3).copy()
    box = A.Aabb(); scale = C.c_float()
    A.check(hip, hip.ngp_sdf_normalize_mesh_host(ptr(verts), C.c_uint64(len(verts)), C.byref(box), C.byref(scale)))
    tn = np.ascontiguousarray(verts.reshape(-1, 3, 3))
    B = 1 << 13
    calls = [3, 1, 1, 5, "iou", 2, 1, "sd", 4, 9]

    def run(ahead):
        cfg = A.sdf_encmlp_config()
        hh = C.c_void_p(); A.check(hip, hip.ngp_encmlp_create(C.byref(cfg), C.c_uint64(1337), C.byref(hh)))
        o = A.default_sdf_options(batch_size=B)
        t = C.c_void_p(); A.check(hip, hip.ngp_sdf_create(hh, ptr(tn), len(tn), box, C.byref(o), C.byref(t)))
        A.check(hip, hip.ngp_sdf_set_batches_ahead(t, ahead))
        seen = []
        for c in calls:
            if c == "iou":
                iou = C.c_double(); A.check(hip, hip.ngp_sdf_iou(t, 1 << 14, C.byref(iou))); seen.append(("iou", iou.value))
            elif c == "sd":  # a ground-truth query of the caller's own points shares the scratch with what is generated ahead
                p = torch.from_numpy(np.random.default_rng(5).uniform(0.2, 0.8, (777, 3)).astype(np.float32)).cuda(); d = torch.zeros(777, device="cuda")
                A.check(hip, hip.ngp_sdf_signed_distance(t, None, C.c_void_p(p.data_ptr()), 777, C.c_void_p(d.data_ptr()))); torch.cuda.synchronize()
                seen.append(("sd", d.cpu().numpy()))
            else:
                A.check(hip, hip.ngp_sdf_train(t, None, c)); torch.cuda.synchronize()
                pp, dp = C.c_void_p(), C.c_void_p(); hip.ngp_sdf_batch_ptrs(t, C.byref(pp), C.byref(dp))
                seen.append(("batch", _dev_read(pp.value, B * 3, np.uint32), _dev_read(dp.value, B, np.uint32)))
        loss = C.c_float(); A.check(hip, hip.ngp_sdf_loss(t, None, C.byref(loss)))
        hip.ngp_sdf_destroy(t); hip.ngp_encmlp_destroy(hh)
        return seen, loss.value

    ref, ref_loss = run(0)
    assert any(np.any(a[2][B // 2:] != b[2][B // 2:]) for a, b in zip([r for r in ref if r[0] == "batch"][:-1], [r for r in ref if r[0] == "batch"][1:]))  # (the batches do differ from one another)
    for ahead in (1, 3, 4, 16):
        got, loss = run(ahead)
        for k, (r, g) in enumerate(zip(ref, got)):
            if r[0] == "batch":
                assert np.array_equal(r[1], g[1]) and np.array_equal(r[2], g[2]), f"{name}: batches_ahead {ahead}: the batch after call {k} ({calls[k]} steps) differs from the serial loop's"
            elif r[0] == "sd":
                assert np.array_equal(r[1], g[1])
            else:
                assert abs(r[1] - g[1]) < 0.02, (ahead, r[1], g[1])
        assert abs(loss - ref_loss) <= 0.05 * abs(ref_loss) + 1e-4, (ahead, loss, ref_loss)
        print(f"{name}: batches_ahead {ahead}: {sum(1 for r in ref if r[0] == 'batch')} batches bit-identical to the serial loop's, loss {loss:.5f} vs {ref_loss:.5f}")
