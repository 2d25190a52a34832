"""Image / SDF primitives' model (HashGrid L=16 F=2 over a 2-D / 3-D position + FullyFusedMLP 2x64): oracle pins on the CPU,
parity of the fused HIP forward kernel against the oracle on the GPU (BASELINE.json configs 0 and 4, forward path)."""
import ctypes as C

import numpy as np
import pytest

import ngp_abi as A
from common import half_to_f32, ptr


def _ora_model(ora, cfg, seed=1337):
    h = C.c_void_p()
    assert ora.ora_encmlp_create(C.byref(cfg), C.c_uint64(seed), C.byref(h)) == 0, ora.ora_last_error()
    n, nm = ora.ora_encmlp_n_params(h), ora.ora_encmlp_n_mlp(h)
    p = np.ctypeslib.as_array(C.cast(ora.ora_encmlp_params_fp(h), C.POINTER(C.c_float)), shape=(n,))
    return h, n, nm, p


def _layout(ora, h, L=16):
    off = np.zeros(L + 1, np.uint32); res = np.zeros(L, np.uint32); sc = np.zeros(L, np.float32)
    ora.ora_encmlp_grid_layout(h, ptr(off), ptr(res), ptr(sc))
    return off, res, sc


def test_image_layout_is_all_dense(ora):
    """with T = 2^19 and a finest level of ~512^2 all 16 levels are dense (~1.42 M grid parameters). The finest
    resolution is 513, not 512: per_level_scale and the level scales are float32 in the reference (testbed.cu:4249, tcnn grid_scale)."""
    cfg = A.image_encmlp_config()
    assert abs(cfg.per_level_scale - 1.25992) < 1e-5
    h, n, nm, _ = _ora_model(ora, cfg)
    off, res, _ = _layout(ora, h)
    assert nm == 64 * 32 + 64 * 64 + 16 * 64
    assert res[0] == 16 and res[-1] in (512, 513)
    for l in range(16):
        assert off[l + 1] - off[l] == (int(res[l]) ** 2 + 7) // 8 * 8 <= 2 ** 19
    assert abs((n - nm) - 1419392) < 8192, n - nm
    ora.ora_encmlp_destroy(h)


def test_sdf_layout(ora):
    """5 dense + 11 hashed levels, ~12.2 M grid parameters."""
    cfg = A.sdf_encmlp_config()
    assert abs(cfg.per_level_scale - 1.3819) < 1e-4
    h, n, nm, _ = _ora_model(ora, cfg)
    off, res, _ = _layout(ora, h)
    sizes = np.diff(off.astype(np.int64))
    dense = [int(res[l]) ** 3 <= 2 ** 19 for l in range(16)]
    assert sum(dense) == 5 and dense[:5] == [True] * 5 and all(sizes[l] == 2 ** 19 for l in range(16) if not dense[l])
    assert abs((n - nm) - 12196240) < 65536, n - nm
    ora.ora_encmlp_destroy(h)


def test_encode_2d_matches_float64_bilinear(ora):
    cfg = A.image_encmlp_config()
    h, n, nm, p = _ora_model(ora, cfg)
    rng = np.random.default_rng(5)
    p[nm:] = rng.uniform(-1, 1, n - nm).astype(np.float32)
    ora.ora_encmlp_sync_half(h)
    off, res, sc = _layout(ora, h)
    table16 = np.empty(n - nm, np.uint16)
    ora.ora_f2h(ptr(np.ascontiguousarray(p[nm:])), ptr(table16), C.c_uint64(n - nm))
    table = half_to_f32(table16).astype(np.float64).reshape(-1, 2)
    N = 257
    uv = rng.random((N, 2), dtype=np.float32)
    out = np.zeros((N, 32), np.uint16)
    assert ora.ora_encmlp_encode(h, ptr(uv), 2, N, ptr(out)) == 0
    got = half_to_f32(out)
    for l in range(16):
        q = np.float64(sc[l]) * uv.astype(np.float64) + 0.5
        g = np.floor(q).astype(np.int64); f = q - g
        acc = np.zeros((N, 2))
        for c in range(4):
            cx, cy = c & 1, (c >> 1) & 1
            w = (f[:, 0] if cx else 1 - f[:, 0]) * (f[:, 1] if cy else 1 - f[:, 1])
            idx = ((g[:, 0] + cx) + (g[:, 1] + cy) * int(res[l])) % int(off[l + 1] - off[l])
            acc += w[:, None] * table[int(off[l]) + idx]
        assert np.abs(got[:, 2 * l:2 * l + 2] - acc).max() < 4e-3, l  # half products / accumulation vs float64
    ora.ora_encmlp_destroy(h)


def test_encode_3d_agrees_with_nerf_grid_oracle(ora):
    """The general-D restatement must reproduce the 3-D NeRF grid encoder (L=8, F=4, T=2^19) bit for bit."""
    from common import OraModel, random_coords
    mc = A.base_model_config(1)
    om = OraModel(ora, mc)
    cfg = A.EncMlpConfig(3, 8, 4, mc.log2_hashmap_size, mc.base_resolution, mc.per_level_scale, 64, 2, 3)
    h, n, nm, p = _ora_model(ora, cfg)
    assert n - nm == om.n - om.n_mlp
    rng = np.random.default_rng(9)
    g = rng.uniform(-1, 1, n - nm).astype(np.float32)
    p[nm:] = g; ora.ora_encmlp_sync_half(h)
    om.params_fp[om.n_mlp:] = g; ora.ora_model_sync_half(om.h)
    c = random_coords(2000, seed=2)
    c[0, :3] = 1.0
    ref = om.encode(c)
    pos = np.ascontiguousarray(c[:, :3])
    out = np.zeros((len(c), 32), np.uint16)
    assert ora.ora_encmlp_encode(h, ptr(pos), 3, len(c), ptr(out)) == 0
    assert np.array_equal(ref, out)
    ora.ora_encmlp_destroy(h)


@pytest.mark.parametrize("which", ["image", "sdf"])
def test_encmlp_gradients_vs_autograd(ora, which):
    """Oracle training step of the image / SDF model (next-round groundwork for ngp_encmlp training): MLP and grid gradients against a
    float64 PyTorch autograd implementation, L2 / MAPE loss gradients against autograd of the loss definitions."""
    import torch
    cfg = A.image_encmlp_config(log2_hashmap_size=14) if which == "image" else A.sdf_encmlp_config(log2_hashmap_size=14)
    D, no = cfg.n_pos_dims, cfg.n_output_dims
    h, n_params, nm, p = _ora_model(ora, cfg)
    rng = np.random.default_rng(2)
    p[:nm] = rng.uniform(-0.3, 0.3, nm).astype(np.float32); p[nm:] = rng.uniform(-1, 1, n_params - nm).astype(np.float32)
    ora.ora_encmlp_sync_half(h)
    off, res, sc = _layout(ora, h)
    n = 400
    x = rng.random((n, D), dtype=np.float32)
    pred16 = np.zeros((n, no), np.uint16)
    assert ora.ora_encmlp_inference(h, ptr(x), D, n, ptr(pred16), no) == 0
    target = rng.normal(size=(n, no)).astype(np.float32) * 0.3
    dl = np.zeros((n, 16), np.uint16)
    loss = ora.ora_encmlp_loss_and_gradient(h, 1 if which == "sdf" else 0, ptr(pred16), no, ptr(target), no, n, 128.0, ptr(dl))
    # --- loss definitions under autograd ---
    pt = torch.tensor(half_to_f32(pred16), dtype=torch.float64, requires_grad=True); tt = torch.tensor(target, dtype=torch.float64)
    lt = ((pt - tt).abs() / (tt.abs() + 0.01)).sum() / (n * no) if which == "sdf" else ((pt - tt) ** 2).sum() / (n * no)
    lt.backward()
    assert abs(loss - float(lt)) < 1e-4 * abs(float(lt))
    gl = half_to_f32(dl)[:, :no]
    assert np.abs(gl - 128.0 * pt.grad.numpy()).max() < 2e-3 * np.abs(128.0 * pt.grad.numpy()).max() + 1e-6
    assert not half_to_f32(dl)[:, no:].any()
    # --- network gradients ---
    assert ora.ora_encmlp_training_step(h, ptr(x), D, n, ptr(dl), 16) == 0
    g = half_to_f32(np.ctypeslib.as_array(C.cast(ora.ora_encmlp_gradients(h), C.POINTER(C.c_uint16)), shape=(n_params,)).copy()).astype(np.float64)
    ph = np.empty(n_params, np.uint16); ora.ora_f2h(ptr(np.ascontiguousarray(p)), ptr(ph), C.c_uint64(n_params))
    P = torch.tensor(half_to_f32(ph), dtype=torch.float64, requires_grad=True)
    X = torch.tensor(x, dtype=torch.float64)
    primes = (1, 2654435761, 805459861)
    feats = []
    for l in range(16):
        hs = int(off[l + 1] - off[l]); r = int(res[l])
        tbl = P[nm + int(off[l]) * 2: nm + int(off[l + 1]) * 2].reshape(-1, 2)
        q = float(np.float32(sc[l])) * X + 0.5
        gi = torch.floor(q).long(); f = q - gi
        acc = 0
        for c in range(1 << D):
            w = torch.ones(n, dtype=torch.float64); ids = []
            for d in range(D):
                bit = (c >> d) & 1
                w = w * (f[:, d] if bit else 1 - f[:, d]); ids.append(gi[:, d] + bit)
            if r ** D > hs:
                idx = 0
                for d in range(D):
                    idx = idx ^ ((ids[d] * primes[d]) & 0xFFFFFFFF)
            else:
                idx = sum(ids[d] * r ** d for d in range(D))
            acc = acc + w[:, None] * tbl[idx % hs]
        feats.append(acc)
    enc = torch.cat(feats, 1)
    W = lambda a, rr, cc: P[a:a + rr * cc].reshape(rr, cc)
    out = torch.relu(torch.relu(enc @ W(0, 64, 32).T) @ W(2048, 64, 64).T) @ W(6144, 16, 64).T
    out.backward(torch.tensor(half_to_f32(dl), dtype=torch.float64))
    gref = P.grad.numpy()
    for name, a, b in (("l1", 0, 2048), ("l2", 2048, 6144), ("l3", 6144, 7168), ("grid", nm, n_params)):
        rel = np.linalg.norm(g[a:b] - gref[a:b]) / np.linalg.norm(gref[a:b])
        assert rel < 3e-2, (which, name, rel)
    ora.ora_encmlp_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("which,n", [("image", 65536), ("image", 77), ("sdf", 50021)])
def test_encmlp_inference_parity(ora, hip, which, n):
    """BASELINE.json config 0 (image, batch 65536 stratified uv) and config 4 (SDF) forward path: fused HIP kernel vs oracle."""
    import torch
    from common import dptr
    cfg = A.image_encmlp_config() if which == "image" else A.sdf_encmlp_config()
    D, n_out = cfg.n_pos_dims, cfg.n_output_dims
    oh, np_, nm, p = _ora_model(ora, cfg)
    hh = C.c_void_p()
    A.check(hip, hip.ngp_encmlp_create(C.byref(cfg), C.c_uint64(1337), C.byref(hh)))
    a, b = C.c_uint64(), C.c_uint64()
    hip.ngp_encmlp_n_params(hh, C.byref(a), C.byref(b))
    assert a.value == np_ and b.value == nm
    init = np.empty(np_, np.float32)
    A.check(hip, hip.ngp_encmlp_get_params_host(hh, ptr(init), C.c_uint64(np_)))
    assert np.array_equal(init, p)  # same initialisation stream as the oracle
    rng = np.random.default_rng(11)
    p[:nm] = rng.uniform(-0.3, 0.3, nm).astype(np.float32)
    p[nm:] = rng.uniform(-1.0, 1.0, np_ - nm).astype(np.float32)
    ora.ora_encmlp_sync_half(oh)
    A.check(hip, hip.ngp_encmlp_set_params_host(hh, ptr(p), C.c_uint64(np_)))
    if which == "image" and n == 65536:  # stratified 256 x 256 uv batch of the image trainer (testbed_image.cu:249-259)
        j = rng.random((256, 256, 2), dtype=np.float32)
        ii, jj = np.meshgrid(np.arange(256, dtype=np.float32), np.arange(256, dtype=np.float32), indexing="ij")
        x = np.ascontiguousarray(((np.stack([jj, ii], -1) + j) / 256.0).reshape(-1, 2).astype(np.float32))
    else:
        x = rng.random((n, D), dtype=np.float32)
        x[0] = 1.0  # upper boundary: dense-level index wrap
    ref = np.zeros((n, n_out), np.uint16)
    assert ora.ora_encmlp_inference(oh, ptr(x), D, n, ptr(ref), n_out) == 0
    xd = torch.from_numpy(x).cuda()
    out = torch.zeros((n, n_out), dtype=torch.int16, device="cuda")
    A.check(hip, hip.ngp_encmlp_inference(hh, None, dptr(xd), D, n, dptr(out), n_out))
    torch.cuda.synchronize()
    got = half_to_f32(out.cpu().numpy().view(np.uint16)); r = half_to_f32(ref)
    err = np.abs(got - r)
    assert (err <= 2e-3 + 1e-2 * np.abs(r)).all(), f"max err {err.max()} ref {r.flat[err.argmax()]}"
    assert np.abs(r).max() > 0.05  # non-trivial outputs
    hip.ngp_encmlp_destroy(hh); ora.ora_encmlp_destroy(oh)


# ------------------------------------------------------------------------------------------------------------------------
# training path (BASELINE configs 0 and 4): HIP kernels vs the oracle
# ------------------------------------------------------------------------------------------------------------------------
def _pair(ora, hip, which, seed=11):
    cfg = A.image_encmlp_config() if which == "image" else A.sdf_encmlp_config()
    oh, np_, nm, p = _ora_model(ora, cfg)
    hh = C.c_void_p()
    A.check(hip, hip.ngp_encmlp_create(C.byref(cfg), C.c_uint64(1337), C.byref(hh)))
    rng = np.random.default_rng(seed)
    p[:nm] = rng.uniform(-0.3, 0.3, nm).astype(np.float32)
    p[nm:] = rng.uniform(-1.0, 1.0, np_ - nm).astype(np.float32)
    ora.ora_encmlp_sync_half(oh)
    A.check(hip, hip.ngp_encmlp_set_params_host(hh, ptr(p), C.c_uint64(np_)))
    return cfg, oh, hh, np_, nm, p, rng


def _dev_read(ptr_value, n, dtype):
    out = np.empty(n, dtype)
    rt = C.CDLL("libamdhip64.so")
    assert rt.hipMemcpy(ptr(out), C.c_void_p(ptr_value), C.c_size_t(out.nbytes), 2) == 0
    return out


def _hip_grads(hip, hh, n_params):
    g = C.c_void_p(); hip.ngp_encmlp_param_ptrs(hh, None, None, None, C.byref(g))
    return _dev_read(g.value, n_params, np.uint16)


def _rel(a, b):
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.gpu
@pytest.mark.parametrize("which,n", [("image", 65536), ("image", 1000), ("sdf", 50021)])
def test_encmlp_training_step_gradients(ora, hip, which, n):
    """Trainer::training_step with an external dL/dy: MLP weight gradients and the encoding's gradient table, HIP vs oracle, per block."""
    import torch
    from common import dptr
    cfg, oh, hh, np_, nm, p, rng = _pair(ora, hip, which)
    D, n_out = cfg.n_pos_dims, cfg.n_output_dims
    x = rng.random((n, D), dtype=np.float32)
    dl = np.zeros((n, 16), np.float16)
    dl[:, :n_out] = (rng.normal(size=(n, n_out)) * (128.0 / n)).astype(np.float16)
    dlu = dl.view(np.uint16)
    assert ora.ora_encmlp_training_step(oh, ptr(x), D, n, ptr(dlu), 16) == 0
    gref = half_to_f32(np.ctypeslib.as_array(C.cast(ora.ora_encmlp_gradients(oh), C.POINTER(C.c_uint16)), shape=(np_,)).copy())
    xd = torch.from_numpy(x).cuda(); dld = torch.from_numpy(dlu.view(np.int16)).cuda()
    A.check(hip, hip.ngp_encmlp_training_step_external(hh, None, dptr(xd), D, n, dptr(dld), 16))
    torch.cuda.synchronize()
    g = half_to_f32(_hip_grads(hip, hh, np_))
    off, res, sc = _layout(ora, oh)
    blocks = {"l1": (0, 2048), "l2": (2048, 6144), "l3": (6144, 6144 + n_out * 64)}
    for l in range(16):
        blocks[f"grid_level_{l}"] = (nm + int(off[l]) * 2, nm + int(off[l + 1]) * 2)
    report = {k: _rel(g[a:b], gref[a:b]) for k, (a, b) in blocks.items()}
    print(which, n, {k: f"{v:.1e}" for k, v in report.items()})
    assert np.isfinite(g).all() and np.all(g[6144 + n_out * 64:7168] == 0)  # padded output rows receive no gradient
    for k, v in report.items():
        assert v < (2e-2 if not k.startswith("grid") else 5e-2), (k, v)  # half atomics: order-dependent rounding
    hip.ngp_encmlp_destroy(hh); ora.ora_encmlp_destroy(oh)


@pytest.mark.gpu
@pytest.mark.parametrize("which,loss", [("image", "L2"), ("sdf", "MAPE")])
def test_encmlp_loss_gradient_and_optimizer(ora, hip, which, loss):
    """Internal-loss training step: the kernel's dL/dy (from ITS half predictions) equals the oracle's loss gradient bit for bit, the loss
    value agrees, and three full steps (training_step + Adam) track the oracle's master parameters."""
    import torch
    from common import dptr
    cfg, oh, hh, np_, nm, p, rng = _pair(ora, hip, which, seed=5)
    D, n_out = cfg.n_pos_dims, cfg.n_output_dims
    n = 8192
    loss_type = A.LOSS_L2 if loss == "L2" else A.LOSS_MAPE
    for step in range(3):
        x = rng.random((n, D), dtype=np.float32)
        tgt = rng.uniform(-0.5, 1.0, (n, n_out)).astype(np.float32)
        xd = torch.from_numpy(x).cuda(); td = torch.from_numpy(tgt).cuda()
        pred = torch.zeros((n, 4), dtype=torch.int16, device="cuda"); ls = torch.zeros(1, dtype=torch.float32, device="cuda")
        A.check(hip, hip.ngp_encmlp_training_step(hh, None, dptr(xd), D, n, dptr(td), n_out, loss_type, C.c_float(128.0), dptr(ls), dptr(pred), 4))
        torch.cuda.synchronize()
        pred_h = pred.cpu().numpy().view(np.uint16)
        # oracle: loss gradient from the HIP predictions (isolates the loss arithmetic), then its own training step on that dL/dy
        dl = np.zeros((n, 16), np.uint16)
        lval = ora.ora_encmlp_loss_and_gradient(oh, int(loss == "MAPE"), ptr(pred_h), 4, ptr(tgt), n_out, n, C.c_float(128.0), ptr(dl))
        assert abs(float(ls.item()) - lval) <= 1e-4 * abs(lval) + 1e-7, (float(ls.item()), lval)
        ref_pred = np.zeros((n, n_out), np.uint16)
        ora.ora_encmlp_inference(oh, ptr(x), D, n, ptr(ref_pred), n_out)
        assert (np.abs(half_to_f32(pred_h[:, :n_out]) - half_to_f32(ref_pred)) <= 2e-3 + 1e-2 * np.abs(half_to_f32(ref_pred))).all()
        assert ora.ora_encmlp_training_step(oh, ptr(x), D, n, ptr(dl), 16) == 0
        gref = np.ctypeslib.as_array(C.cast(ora.ora_encmlp_gradients(oh), C.POINTER(C.c_uint16)), shape=(np_,)).copy()
        g = _hip_grads(hip, hh, np_)
        assert _rel(half_to_f32(g[:nm]), half_to_f32(gref[:nm])) < 2e-2 and _rel(half_to_f32(g[nm:]), half_to_f32(gref[nm:])) < 5e-2
        # Adam on the ORACLE's gradient on both sides (isolates the optimizer arithmetic)
        gp = C.c_void_p(); hip.ngp_encmlp_param_ptrs(hh, None, None, None, C.byref(gp))
        rt = C.CDLL("libamdhip64.so")
        assert rt.hipMemcpy(gp, ptr(gref), C.c_size_t(gref.nbytes), 1) == 0
        assert ora.ora_encmlp_optimizer_step(oh, C.c_float(128.0)) == 0
        A.check(hip, hip.ngp_encmlp_optimizer_step(hh, None, C.c_float(128.0)))
        torch.cuda.synchronize()
        pm = np.empty(np_, np.float32)
        A.check(hip, hip.ngp_encmlp_get_params_host(hh, ptr(pm), C.c_uint64(np_)))
        assert np.allclose(pm, p, rtol=2e-5, atol=1e-8), np.abs(pm - p).max()  # p aliases the oracle's master parameters
    assert hip.ngp_encmlp_step(hh) == 3
    hip.ngp_encmlp_destroy(hh); ora.ora_encmlp_destroy(oh)


def _test_image(res=256):
    """procedural RGBA float image with edges and gradients (linear colours)"""
    y, x = np.mgrid[0:res, 0:res].astype(np.float32) / res
    img = np.zeros((res, res, 4), np.float32)
    img[..., 0] = 0.5 + 0.5 * np.sin(20 * x) * np.cos(14 * y)
    img[..., 1] = ((x - 0.5) ** 2 + (y - 0.4) ** 2 < 0.08).astype(np.float32) * 0.8 + 0.1
    img[..., 2] = x * y
    img[..., 3] = 1.0
    return np.ascontiguousarray(img)


@pytest.mark.gpu
@pytest.mark.parametrize("snap,linear", [(1, 0), (0, 0), (1, 1), (0, 1)])
def test_image_batch_matches_oracle(ora, hip, snap, linear):
    """train_image's batch: stratified uv positions (bit-exact) and targets (bit-exact for linear colours, 1e-6 through powf otherwise)"""
    import torch
    img = _test_image(200)  # non power of two: exercises the clamps
    cfg = A.image_encmlp_config(image_resolution=200)
    hh = C.c_void_p(); A.check(hip, hip.ngp_encmlp_create(C.byref(cfg), C.c_uint64(1337), C.byref(hh)))
    for batch, strat in ((1 << 14, 1), (1 << 13, 1), (5024, 1), (4096, 0)):
        o = A.default_image_options(snap_to_pixel_centers=snap, linear_colors=linear, stratified=strat, batch_size=batch)
        t = C.c_void_p(); A.check(hip, hip.ngp_image_create(hh, ptr(img), A.IMAGE_FLOAT, 200, 200, C.byref(o), C.byref(t)))
        rng = A.Pcg32(); ora.ora_pcg32_seed(C.byref(rng), C.c_uint64(1337), C.c_uint64(1))
        for step in range(2):
            A.check(hip, hip.ngp_image_train(t, None, 1))
            torch.cuda.synchronize()
            pp, tp = C.c_void_p(), C.c_void_p(); hip.ngp_image_batch_ptrs(t, C.byref(pp), C.byref(tp))
            pos, tgt = _dev_read(pp.value, batch * 2, np.float32), _dev_read(tp.value, batch * 3, np.float32)
            rpos, rtgt = np.zeros(batch * 2, np.float32), np.zeros(batch * 3, np.float32)
            ora.ora_image_generate_batch(ptr(img), 200, 200, batch, rng, strat, snap, linear, ptr(rpos), ptr(rtgt))
            ora.ora_pcg32_advance(C.byref(rng), C.c_int64(batch * 2))
            assert np.array_equal(pos.view(np.uint32), rpos.view(np.uint32)), (batch, strat, step)
            if linear:
                assert np.array_equal(tgt.view(np.uint32), rtgt.view(np.uint32))
            else:
                assert np.allclose(tgt, rtgt, rtol=0, atol=2e-6)
            assert 0 <= pos.min() and pos.max() <= 1
        hip.ngp_image_destroy(t)
    hip.ngp_encmlp_destroy(hh)


@pytest.mark.gpu
def test_image_trainer_learns_the_image(hip):
    """train_image + compute_image_mse end to end (BASELINE config 0 shape: L = 16, F = 2, T = 2^19, MLP 2x64, batch 65536): PSNR after 1000 steps.
    albert.exr (the reference's image, staged under _ref_data/) when present, else a procedural 256^2 image."""
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exr = os.path.join(root, "_ref_data", "data", "image", "albert.exr")
    if os.path.exists(exr):
        import pyngp
        img = np.ascontiguousarray(pyngp.read_exr(exr)); name = "albert.exr"
    else:
        img = _test_image(256); name = "procedural"
    h, w = img.shape[:2]
    cfg = A.image_encmlp_config(image_resolution=max(w, h))
    hh = C.c_void_p(); A.check(hip, hip.ngp_encmlp_create(C.byref(cfg), C.c_uint64(1337), C.byref(hh)))
    o = A.default_image_options()
    t = C.c_void_p(); A.check(hip, hip.ngp_image_create(hh, ptr(img), A.IMAGE_FLOAT, w, h, C.byref(o), C.byref(t)))
    mse0 = C.c_float(); A.check(hip, hip.ngp_image_mse(t, 0, C.byref(mse0)))
    A.check(hip, hip.ngp_image_train(t, None, 20)); torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    A.check(hip, hip.ngp_image_train(t, None, 980)); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mse, mseq, loss = C.c_float(), C.c_float(), C.c_float()
    A.check(hip, hip.ngp_image_mse(t, 0, C.byref(mse))); A.check(hip, hip.ngp_image_mse(t, 1, C.byref(mseq))); A.check(hip, hip.ngp_image_loss(t, None, C.byref(loss)))
    psnr = -10 * np.log10(mse.value)
    print(f"{name} {w}x{h}: mse {mse0.value:.4f} -> {mse.value:.3e} (PSNR {psnr:.2f} dB, quantised {-10 * np.log10(mseq.value):.2f} dB), last batch loss {loss.value:.3e}, "
          f"{980 * o.batch_size / dt / 1e6:.1f} M samples/s ({dt / 980 * 1e3:.3f} ms/step)")
    assert np.isfinite(mse.value) and mse.value < 0.05 * mse0.value and psnr > 28.0
    hip.ngp_image_destroy(t); hip.ngp_encmlp_destroy(hh)
