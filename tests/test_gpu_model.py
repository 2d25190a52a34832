"""GPU parity: fused hash-grid + MLP kernels (through the C-ABI) vs the CPU oracle on the same seeded inputs.

Tolerances (fp16 data path, fp32 accumulation inside the matrix products):
  * encoding features: bit-exact (same fused half FMA chain, same corner order);
  * network outputs: |delta| <= 2e-3 + 1e-2*|ref|  (different fp32 summation order inside the MFMA, then one
    half rounding per layer);
  * gradients: compared as relative L2 error per parameter block (half atomics are order dependent).
"""
import ctypes as C

import numpy as np
import pytest

import ngp_abi as A
from common import HipModel, OraModel, dptr, half_to_f32, ptr, random_coords

pytestmark = pytest.mark.gpu


def _models(ora, hip, **cfg_kw):
    cfg = A.base_model_config(1, **cfg_kw)
    om = OraModel(ora, cfg)
    hm = HipModel(hip, cfg)
    # make the two models bit-identical (initialisation itself is compared in test_init_parity)
    rng = np.random.default_rng(7)
    p = om.params_fp
    p[: om.n_mlp] = rng.uniform(-0.3, 0.3, om.n_mlp).astype(np.float32)
    p[om.n_mlp:] = rng.uniform(-1.0, 1.0, om.n - om.n_mlp).astype(np.float32)  # trained-like feature magnitudes
    ora.ora_model_sync_half(om.h)
    hm.set_params(p)
    return cfg, om, hm


def test_init_parity(ora, hip):
    import torch
    cfg = A.base_model_config(1)
    om = OraModel(ora, cfg, seed=1337)
    hm = HipModel(hip, cfg, seed=1337)
    assert om.n == hm.n == 10240 + 2920448 * 4
    assert np.array_equal(om.params_fp, hm.read("master", torch))
    assert np.array_equal(om.params, hm.read("params", torch))


@pytest.mark.parametrize("n,coherent", [(1, False), (31, False), (4096, False), (100003, True)])
def test_encode_bit_exact(ora, hip, n, coherent):
    import torch
    cfg, om, hm = _models(ora, hip)
    c = random_coords(n, seed=n, ray_coherent=coherent)
    c[0, 0:3] = (1.0, 1.0, 1.0) if n > 1 else c[0, 0:3]  # upper boundary: exercises the dense-level index wrap
    ref = om.encode(c)
    cd = torch.from_numpy(c).cuda()
    out = torch.zeros((n, 32), dtype=torch.int16, device="cuda")
    A.check(hip, hip.ngp_model_encode(hm.h, None, dptr(cd), 7, n, dptr(out)))
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint16)
    bad = np.argwhere(ref != got)
    msg = [(int(i), int(f), float(half_to_f32(ref[i, f])), float(half_to_f32(got[i, f])), c[i, :3].tolist()) for i, f in bad[:8]]
    assert len(bad) == 0, f"{len(bad)} of {ref.size} encoded halfs differ: {msg}"


@pytest.mark.parametrize("n", [1, 33, 64, 5000, 70001])
def test_inference_parity(ora, hip, n):
    import torch
    cfg, om, hm = _models(ora, hip)
    c = random_coords(n, seed=3 * n + 1, ray_coherent=(n > 1000))
    ref = half_to_f32(om.inference(c))
    cd = torch.from_numpy(c).cuda()
    out = torch.zeros((n, 4), dtype=torch.int16, device="cuda")
    A.check(hip, hip.ngp_model_inference(hm.h, None, dptr(cd), 7, n, None, dptr(out), 4, 0))
    torch.cuda.synchronize()
    got = half_to_f32(out.cpu().numpy().view(np.uint16))
    err = np.abs(got - ref)
    tol = 2e-3 + 1e-2 * np.abs(ref)
    assert (err <= tol).all(), f"max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)} ref {ref.flat[err.argmax()]}"


def test_inference_device_count(ora, hip):
    """n_ptr bounds the batch on the device (no host read-back of the K1 counter)."""
    import torch
    cfg, om, hm = _models(ora, hip)
    n_max, n_real = 4096, 1000
    c = random_coords(n_max, seed=5)
    cd = torch.from_numpy(c).cuda()
    out = torch.full((n_max, 4), 0x7bff, dtype=torch.int16, device="cuda")
    cnt = torch.tensor([n_real], dtype=torch.int32, device="cuda")
    A.check(hip, hip.ngp_model_inference(hm.h, None, dptr(cd), 7, n_max, dptr(cnt), dptr(out), 4, 0))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert (o[n_real:] == 0x7bff).all() and not (o[:n_real] == 0x7bff).all()


def test_density_parity(ora, hip):
    import torch
    cfg, om, hm = _models(ora, hip)
    n = 20011
    pos = np.ascontiguousarray(random_coords(n, seed=11)[:, :3])
    ref = half_to_f32(om.density(pos))
    pd = torch.from_numpy(pos).cuda()
    out = torch.zeros((n,), dtype=torch.int16, device="cuda")
    A.check(hip, hip.ngp_model_density(hm.h, None, dptr(pd), 3, n, dptr(out), 1, 0))
    torch.cuda.synchronize()
    got = half_to_f32(out.cpu().numpy().view(np.uint16))
    err = np.abs(got - ref)
    assert (err <= 2e-3 + 1e-2 * np.abs(ref)).all(), err.max()


def _rel_l2(a, b):
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-20))


@pytest.mark.parametrize("n", [256, 8192])
def test_training_step_gradients(ora, hip, n):
    import torch
    cfg, om, hm = _models(ora, hip)
    c = random_coords(n, seed=21, ray_coherent=True)
    rng = np.random.default_rng(5)
    dl = (rng.normal(size=(n, 4)) * (128.0 / n)).astype(np.float16).view(np.uint16)
    om.training_step(c, dl)
    gref = half_to_f32(om.grads.copy())
    om.training_step(c, dl, grid_sum_mode=2)
    gtrue = half_to_f32(om.grads.copy())
    cd = torch.from_numpy(c).cuda(); dld = torch.from_numpy(dl.view(np.int16)).cuda()
    A.check(hip, hip.ngp_model_training_step(hm.h, None, dptr(cd), 7, n, dptr(dld), 4))
    torch.cuda.synchronize()
    ggot = half_to_f32(hm.read("grads", torch))
    blocks = {"density_l1": (0, 2048), "density_l2": (2048, 3072), "rgb_l1": (3072, 5120), "rgb_l2": (5120, 9216), "rgb_l3": (9216, 10240)}
    offs = (C.c_uint32 * 9)(); res = (C.c_uint32 * 8)(); sc = (C.c_float * 8)()
    hip.ngp_model_grid_layout(hm.h, offs, res, sc)
    for l in range(8):
        blocks[f"grid_level_{l}"] = (10240 + offs[l] * 4, 10240 + offs[l + 1] * 4)
    report = {k: _rel_l2(ggot[a:b], gref[a:b]) for k, (a, b) in blocks.items()}
    report_true = {k: _rel_l2(ggot[a:b], gtrue[a:b]) for k, (a, b) in blocks.items() if k.startswith("grid")}
    noise = {k: _rel_l2(gref[a:b], gtrue[a:b]) for k, (a, b) in blocks.items() if k.startswith("grid")}
    print(n, "device vs reference-order oracle", {k: f"{v:.2e}" for k, v in report.items()})
    print(n, "device vs unrounded-sum oracle  ", {k: f"{v:.2e}" for k, v in report_true.items()})
    print(n, "reference-order vs unrounded-sum", {k: f"{v:.2e}" for k, v in noise.items()})
    # rgb_l3 rows 3..15 receive no gradient
    assert np.all(ggot[9216 + 3 * 64:10240] == 0)
    # MLP blocks: fp32 accumulation on both sides (MFMA order vs sequential), half activations and half dL/d(hidden) on both: measured 3e-5 .. 6e-5 at n = 256 and
    # 1e-4 .. 1.6e-3 at n = 8192 (this test's parameters are the INITIAL ones: table values +-1e-4, so the density network's activations are half subnormals and one
    # different rounding of a 1e-5 activation is a per-cent change of that term; the trained-state batch of the test below agrees to <= 2e-4).
    # Grid: the reference adds (half)(dL/d(enc) * weight) one by one in half (one rounding per contribution: small addends are swamped); the device sums the half
    # contributions exactly and merges runs of samples in one cell in fp32 before the half rounding.  Measured against the UNROUNDED sums (the gradient the half
    # dL/d(enc) implies) the device must not be farther away than the reference-order result is.  Bounds ~2 x measured (round 4, profiles/r04_pytest_gpu.log).
    for k, v in report.items():
        if not k.startswith("grid"):
            assert v < (2e-4 if n <= 256 else 4e-3), (k, v, report)
        else:
            # (here dL/d(enc) itself differs by the MLP backward's ~1e-3, which bounds what the grid levels can agree to: measured <= 4.4e-4 at n = 256, <= 1.3e-3 at
            # n = 8192; the statement "at least as close to the unrounded sums as the reference-order result" is made on the trained-state batch of the next test)
            assert report_true[k] < 3e-3 and v < noise[k] + report_true[k] + 1e-6, (k, v, report_true[k], noise[k])
    # sparsity pattern of the hash-grid gradient must match exactly where the reference value is not tiny
    big = np.abs(gref[10240:]) > 1e-4
    assert np.all(ggot[10240:][big] != 0)


GRID_TRUE_TOL = 3e-3  # rel-L2 of the coarse levels (0 - 4), device vs the unrounded-sum oracle: measured 4.0e-4 .. 1.4e-3 (the MLP backward's rounding of dL/d(enc) and one half rounding
                      # per merged record); on the fine levels BOTH paths are bounded by the half rounding of the individual (often subnormal) contributions: 2.8e-3 .. 6.2e-3


def test_hashed_level_gradients_are_reproducible_and_exactly_summed(ora, hip):
    """The hashed levels' gradients go through per-chunk record lists and 64-bit fixed-point LDS accumulators: sums of halfs are
    exact there, so (i) two runs give bit-identical results whatever order the records arrive in, and (ii) the result equals the
    atomics path (flag 2048, half-precision adds in arrival order) up to that path's rounding."""
    import torch
    cfg, om, hm = _models(ora, hip)
    n = 20000  # not a multiple of the 512-sample bin blocks
    c = random_coords(n, seed=77, ray_coherent=True)
    rng = np.random.default_rng(3)
    dl = (rng.normal(size=(n, 4)) * (128.0 / n)).astype(np.float16).view(np.uint16)
    cd = torch.from_numpy(c).cuda(); dld = torch.from_numpy(dl.view(np.int16)).cuda()
    offs = (C.c_uint32 * 9)(); res = (C.c_uint32 * 8)(); sc = (C.c_float * 8)()
    hip.ngp_model_grid_layout(hm.h, offs, res, sc)
    hashed = [l for l in range(8) if int(res[l]) ** 3 > offs[l + 1] - offs[l]]
    assert hashed == [3, 4, 5, 6, 7]
    runs = []
    for flags in (0, 0, 2048):
        hip.ngp_debug_set_flags(flags)
        try:
            A.check(hip, hip.ngp_model_training_step(hm.h, None, dptr(cd), 7, n, dptr(dld), 4))
            torch.cuda.synchronize()
        finally:
            hip.ngp_debug_set_flags(0)
        runs.append(hm.read("grads", torch).copy())
    a0, a1, atom = runs
    for l in hashed:
        lo, hi_ = 10240 + offs[l] * 4, 10240 + offs[l + 1] * 4
        assert np.array_equal(a0[lo:hi_], a1[lo:hi_]), f"level {l}: binned gradients differ between two runs"
        assert _rel_l2(half_to_f32(a0[lo:hi_]), half_to_f32(atom[lo:hi_])) < 2e-2, l
        assert np.count_nonzero(a0[lo:hi_]) > 0


@pytest.mark.parametrize("ema_full_precision", [0, 1])
def test_optimizer_step_parity(ora, hip, ema_full_precision):
    """Adam + the two EMA kernels of tcnn's EmaOptimizer: the default half-precision state (the inference buffer itself, fed with the half weights) and
    "full_precision" (fp32 state fed with the fp32 masters); the half mode must agree to the BIT with the oracle given equal weights (same fp32 expression, one rounding)."""
    import torch
    cfg, om, hm = _models(ora, hip, ema_full_precision=ema_full_precision)
    n = 4096
    c = random_coords(n, seed=33, ray_coherent=True)
    rng = np.random.default_rng(9)
    for step in range(3):
        dl = (rng.normal(size=(n, 4)) * (128.0 / n)).astype(np.float16).view(np.uint16)
        om.training_step(c, dl)
        cd = torch.from_numpy(c).cuda(); dld = torch.from_numpy(dl.view(np.int16)).cuda()
        A.check(hip, hip.ngp_model_training_step(hm.h, None, dptr(cd), 7, n, dptr(dld), 4))
        # feed the ORACLE gradient to the HIP optimizer so that this test isolates Adam/EMA arithmetic
        m_, p_, i_, g_ = hm.ptrs()
        g = om.grads.copy()
        rt = C.CDLL("libamdhip64.so")
        torch.cuda.synchronize()
        assert rt.hipMemcpy(C.c_void_p(g_), ptr(g), C.c_size_t(g.nbytes), 1) == 0
        ora.ora_model_optimizer_step(om.h, C.c_float(128.0))
        A.check(hip, hip.ngp_model_optimizer_step(hm.h, None, C.c_float(128.0)))
        torch.cuda.synchronize()
        pm = hm.read("master", torch)
        ref = om.params_fp
        # powf/sqrtf differ in the last ulp between glibc and the device library: allow 1e-6 relative on the update
        assert np.allclose(pm, ref, rtol=2e-5, atol=1e-8), np.abs(pm - ref).max()
        inf = half_to_f32(hm.read("inference", torch)); rinf = half_to_f32(om.params_inf)
        assert np.allclose(inf, rinf, rtol=2e-3, atol=1e-6)
        if step == 0 and not ema_full_precision: assert np.array_equal(hm.read("inference", torch), hm.read("params", torch))  # debias_old = 0: the first EMA of half weights is the weights
        # keep both models identical for the next iteration
        hm.set_params(ref.copy()) if False else None
    assert hip.ngp_model_step(hm.h) == 3


def _captured_batch(hip, B, steps):
    """one compacted training batch of the NeRF trainer (K1-marched, K3-compacted, K4-padded coordinates and loss gradients) and the
    trained parameters it was computed with"""
    import torch
    from common import host_meta, make_small_dataset
    imgs, xforms, meta = make_small_dataset(12, 96)
    M, X = host_meta(imgs, xforms, meta)
    cfg = A.base_model_config(1)
    hm = HipModel(hip, cfg)
    t = C.c_void_p()
    opts = A.default_nerf_options(1, target_batch_size=B)
    A.check(hip, hip.ngp_nerf_create(hm.h, C.byref(opts), A.scene_aabb(1), C.byref(t)))
    pix = (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
    A.check(hip, hip.ngp_nerf_set_dataset_host(t, len(imgs), M, X, pix))
    A.check(hip, hip.ngp_nerf_train(t, None, steps))
    A.check(hip, hip.ngp_nerf_train_prep(t, None))
    A.check(hip, hip.ngp_nerf_train_forward(t, None))  # K1..K4: the batch of step `steps`, not yet consumed
    torch.cuda.synchronize()
    ri, rays, ns, co, mo, cc, dl, cnt = (C.c_void_p() for _ in range(8))
    A.check(hip, hip.ngp_nerf_scratch_ptrs(t, C.byref(ri), C.byref(rays), C.byref(ns), C.byref(co), C.byref(mo), C.byref(cc), C.byref(dl), C.byref(cnt)))
    coords = np.empty((B, 7), np.float32); dloss = np.empty((B, 4), np.uint16)
    rt = C.CDLL("libamdhip64.so")
    assert rt.hipMemcpy(ptr(coords), cc, C.c_size_t(coords.nbytes), 2) == 0 and rt.hipMemcpy(ptr(dloss), dl, C.c_size_t(dloss.nbytes), 2) == 0
    params = np.empty(hm.n, np.float32)
    A.check(hip, hip.ngp_model_get_params_host(hm.h, ptr(params), C.c_uint64(params.size)))
    st = A.NerfStats(); A.check(hip, hip.ngp_nerf_get_stats(t, None, C.byref(st)))
    hip.ngp_nerf_destroy(t)
    return cfg, hm, params, coords, dloss, st


def test_training_step_gradients_full_batch_and_bin_layouts(ora, hip):
    """Gradient parity at the benchmark's batch size (B = 2^18 samples) on a batch captured from the NeRF trainer: HIP vs oracle per
    parameter block, for every layout of the hashed levels' binned scatter (chunk 2^11 / 2^12, one block per chunk or per (chunk,
    feature pair)), with the list capacity forced small (the overflow path: most records then take global atomics), and for the
    atomics-only path.  The binned layouts sum the same multiset of half contributions exactly => bit-identical among themselves."""
    import torch
    B = 1 << 18
    cfg, hm, params, coords, dloss, st = _captured_batch(hip, B, 48)
    assert st.measured_batch_size > 0 and np.isfinite(coords).all()
    om = OraModel(ora, cfg)
    om.params_fp[:] = params
    ora.ora_model_sync_half(om.h)
    om.training_step(coords, dloss)
    gref = half_to_f32(om.grads.copy())
    om.training_step(coords, dloss, grid_sum_mode=2)
    gtrue = half_to_f32(om.grads.copy())
    cd = torch.from_numpy(coords).cuda(); dld = torch.from_numpy(dloss.view(np.int16)).cuda()
    offs = (C.c_uint32 * 9)(); res = (C.c_uint32 * 8)(); sc = (C.c_float * 8)()
    hip.ngp_model_grid_layout(hm.h, offs, res, sc)
    blocks = {"density_l1": (0, 2048), "density_l2": (2048, 3072), "rgb_l1": (3072, 5120), "rgb_l2": (5120, 9216), "rgb_l3": (9216, 9216 + 3 * 64)}
    for l in range(8):
        blocks[f"grid_level_{l}"] = (10240 + offs[l] * 4, 10240 + offs[l + 1] * 4)
    noise = {k: _rel_l2(gref[a:b], gtrue[a:b]) for k, (a, b) in blocks.items() if k.startswith("grid")}
    print("reference-order oracle vs unrounded-sum oracle (the reference's own half-accumulation error)", {k: f"{v:.2e}" for k, v in noise.items()})
    hashed = [l for l in range(8) if int(res[l]) ** 3 > offs[l + 1] - offs[l]]
    variants = [("chunk12", 12, 0, 0, 0), ("chunk12_split", 12, 1, 0, 0), ("chunk11", 11, 0, 0, 0), ("chunk11_split", 11, 1, 0, 0),
                ("chunk12_overflow", 12, 0, 2048, 0), ("chunk11_split_overflow", 11, 1, 1024, 0), ("atomics_only", 12, 0, 0, 2048),
                # ablation variants of ngp_kernels.hpp: run merging in k_grad_bin, dense levels through k_grad_dense, one-role weight-gradient kernel
                ("bin_no_hashed_merge", 12, 0, 0, 65536), ("w_single_role", 12, 0, 0, 32768),
                # dense levels as half atomics from T1 (rounds 1-2a) / from k_grad_dense instead of through the bin lists
                ("t1_dense_atomics", 12, 0, 0, 8388608), ("t1_dense_atomics_chunk11", 11, 0, 0, 8388608), ("dense_external", 12, 0, 0, 262144 | 8388608)]
    got = {}
    try:
        for name, cl2, split, cap, flags in variants:
            A.check(hip, hip.ngp_debug_set_bin_params(cl2, split, cap)); hip.ngp_debug_set_flags(flags)
            A.check(hip, hip.ngp_model_training_step(hm.h, None, dptr(cd), 7, B, dptr(dld), 4))
            torch.cuda.synchronize()
            g = hm.read("grads", torch).copy()
            got[name] = g
            gf = half_to_f32(g)
            report = {k: _rel_l2(gf[a:b], gref[a:b]) for k, (a, b) in blocks.items()}
            report_true = {k: _rel_l2(gf[a:b], gtrue[a:b]) for k, (a, b) in blocks.items() if k.startswith("grid")}
            print(name, {k: f"{v:.2e}" for k, v in report.items()})
            print(name, "vs unrounded sums", {k: f"{v:.2e}" for k, v in report_true.items()})
            assert np.isfinite(gf).all()
            # which levels this variant sums exactly (fixed-point lists) and which through half atomics (order dependent, like the reference)
            atomic_levels = set(range(8)) if name == "atomics_only" else set(l for l in range(8) if l not in hashed) if ((flags & 8388608) or split) else set()
            if "overflow" in name:
                atomic_levels = set(range(8))   # most records take the fallback atomics
            for k, v in report.items():
                if not k.startswith("grid"):
                    assert v < 1e-3, (name, k, v)   # measured <= 2e-4
                    continue
                l = int(k[-1])
                assert v < noise[k] + report_true[k] + 1e-6, (name, k, v, noise[k], report_true[k])   # triangle inequality: nothing but the two known terms
                if l not in atomic_levels:
                    # the statement of this test: at EVERY level the exactly summed device gradient is at least as close to the unrounded sums as the reference's chain of
                    # half adds is (measured: level 0 4.0e-4 vs 1.55e-2, level 2 6.2e-4 vs 2.3e-3, level 4 1.4e-3 vs 1.9e-3, levels 5 - 7 equal to three digits)
                    assert report_true[k] <= 1.05 * noise[k] + 2e-5, ("exact device sums must be as close to the unrounded sums as the reference-order oracle is", name, k, report_true[k], noise[k])
                    if l <= 4:
                        assert report_true[k] < GRID_TRUE_TOL, (name, k, report_true[k])
                else:
                    assert report_true[k] < 2.0 * noise[k] + GRID_TRUE_TOL, (name, k, report_true[k], noise[k])    # half atomics: the reference's own kind of error
    finally:
        A.check(hip, hip.ngp_debug_set_bin_params(11, 0, 0)); hip.ngp_debug_set_flags(0)  # (11: the library default since round 6)
    # Bit-identical sums within one instantiation of T1 (the layouts with one block per chunk run the T1 without scatter code, the split ones
    # the T1 that issues the dense levels' atomics: the compiler contracts the MLP's fp operations differently in the two, so dL/d(enc)
    # differs in the last bit between them); in the one-block-per-chunk layouts the DENSE levels are exact sums as well.
    for l in range(8):
        lo, hi_ = blocks[f"grid_level_{l}"]
        assert np.array_equal(got["chunk12"][lo:hi_], got["chunk11"][lo:hi_]), ("chunk11", l)
        if l in hashed:
            assert np.array_equal(got["chunk12_split"][lo:hi_], got["chunk11_split"][lo:hi_]), ("chunk11_split", l)
            assert np.array_equal(got["t1_dense_atomics"][lo:hi_], got["t1_dense_atomics_chunk11"][lo:hi_]), ("t1_dense_atomics_chunk11", l)
