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
