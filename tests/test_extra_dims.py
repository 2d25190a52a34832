"""Extra (latent / light-direction) dims of the NeRF network (VERDICT r3 Missing 1).
Reference: nerf_network.h:81-95 (the dir encoding takes n_dir_dims + n_extra_dims inputs: Composite = SphericalHarmonics over the direction, Identity over the rest,
padded to the colour network's alignment), testbed_nerf.cu:718-744, 833 (K1 copies the image's extra dims behind every NerfCoordinate), :1293-1330
(compute_extra_dims_gradient_train_nerf), :2860-2878 + adam_optimizer.h:27-47 (one VarAdamOptimizer per image), :3656-3707 (reset / rendering extra dims),
nerf_loader.h:82-87 (n_extra_dims = 3 light-direction dims + n_extra_learnable_dims).
CPU tests pin the oracle's statement of the layout; GPU tests compare the HIP kernels with it through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

import ngp_abi as A
from common import HipModel, OraModel, dptr, half_to_f32, host_meta, make_small_dataset, ptr, random_coords

N_BASE_MLP = 64 * 32 + 16 * 64 + 64 * 32 + 64 * 64 + 16 * 64  # configs/nerf/base.json: 10,240


def _coords(n, n_extra, seed, ray_coherent=True):
    c7 = random_coords(n, seed=seed, ray_coherent=ray_coherent)
    rng = np.random.default_rng(seed + 1000)
    # extra dims are per IMAGE in the trainer: constant over runs of samples here (32 samples = one "ray")
    per_ray = rng.uniform(-1, 1, ((n + 31) // 32, n_extra)).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([c7, per_ray[np.arange(n) // 32]], axis=1))


def _fill(om, seed=7):
    rng = np.random.default_rng(seed)
    p = om.params_fp
    p[: om.n_mlp] = rng.uniform(-0.3, 0.3, om.n_mlp).astype(np.float32)
    p[om.n_mlp:] = rng.uniform(-1.0, 1.0, om.n - om.n_mlp).astype(np.float32)
    return p


def _ora_inference(ora, om, c):
    out = np.zeros((c.shape[0], 4), np.uint16)
    ora.ora_model_inference(om.h, ptr(c), c.shape[1], c.shape[0], ptr(out), 4, 0)
    return out


# ---------------------------------------------------------------- CPU: the oracle's layout
@pytest.mark.parametrize("n_extra", [3, 16])
def test_oracle_layout_of_the_wider_first_colour_layer(ora, n_extra):
    """The colour network's first layer becomes 64 x 48 ([density output 16 | SH 16 | extra dims + padding 16], nerf_network.h:84-95); with its extra
    columns zeroed the network is the plain base.json network on the same remaining weights, whatever the extra dims are."""
    cfg_x = A.base_model_config(1, n_extra_dims=n_extra); cfg_0 = A.base_model_config(1)
    ox, o0 = OraModel(ora, cfg_x), OraModel(ora, cfg_0)
    assert ox.n_mlp == N_BASE_MLP + 64 * 16 and o0.n_mlp == N_BASE_MLP and ox.n - ox.n_mlp == o0.n - o0.n_mlp
    p0 = _fill(o0)
    px = ox.params_fp
    px[:3072] = p0[:3072]                                                    # density network
    w1 = np.zeros((64, 48), np.float32); w1[:, :32] = p0[3072:5120].reshape(64, 32)
    px[3072:3072 + 64 * 48] = w1.ravel()                                     # first colour layer, extra columns zero
    px[3072 + 64 * 48: ox.n_mlp] = p0[5120: o0.n_mlp]                        # the layers behind it
    px[ox.n_mlp:] = p0[o0.n_mlp:]                                            # hash grid
    ora.ora_model_sync_half(ox.h); ora.ora_model_sync_half(o0.h)
    c = _coords(2000, n_extra, 3)
    a = _ora_inference(ora, ox, c)
    b = _ora_inference(ora, o0, np.ascontiguousarray(c[:, :7]))
    assert np.array_equal(a, b)
    # ... and with them set, the padding columns (inputs = 1) act as a bias: a change of an extra dim changes the colour, never the density
    px[3072:3072 + 64 * 48] = np.random.default_rng(1).uniform(-0.3, 0.3, 64 * 48).astype(np.float32)
    ora.ora_model_sync_half(ox.h)
    c2 = c.copy(); c2[:, 7] += 0.5
    a1, a2 = _ora_inference(ora, ox, c), _ora_inference(ora, ox, c2)
    assert np.array_equal(a1[:, 3], a2[:, 3]) and (a1[:, :3] != a2[:, :3]).any()


def test_oracle_var_adam_and_gradient_reduction(ora):
    """the two small restatements the GPU tests check against: adam_optimizer.h:37-47 in numpy float32, testbed_nerf.cu:1293-1330 as a plain loop"""
    rng = np.random.default_rng(0)
    n = 40
    var = rng.normal(size=n).astype(np.float32); g = (rng.normal(size=n) * 128).astype(np.float32)
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    var0 = var.copy()
    ora.ora_var_adam_step(n, ptr(var), ptr(g), ptr(m), ptr(v), 1, C.c_float(1e-2), C.c_float(128.0))
    gs = g / np.float32(128.0)
    assert np.allclose(m, 0.1 * gs, rtol=1e-6) and np.allclose(v, 0.01 * gs * gs, rtol=1e-5)
    # first step of Adam: the debiased update is lr * sign(g) (up to epsilon)
    assert np.allclose(var0 - var, 1e-2 * np.sign(gs), rtol=1e-3)
    n_rays_total, n_img, n_extra = 64, 4, 3
    ray_idx = rng.permutation(n_rays_total)[:20].astype(np.uint32)
    counts = rng.integers(0, 6, 20).astype(np.uint32); base = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint32)
    ns = np.ascontiguousarray(np.stack([counts, base], 1))
    dx = rng.normal(size=(int(counts.sum()), n_extra)).astype(np.float32)
    out = np.zeros((n_img, n_extra), np.float32)
    ora.ora_extra_dims_gradient(n_rays_total, 20, ptr(out), n_extra, n_img, ptr(ray_idx), ptr(ns), ptr(dx))
    ref = np.zeros_like(out)
    for r in range(20):
        img = (int(ray_idx[r]) * n_img // n_rays_total) % n_img  # image_idx, nerf_device.cuh:593-599
        ref[img] += dx[base[r]:base[r] + counts[r]].sum(0)
    assert np.allclose(out, ref, atol=1e-5)


# ---------------------------------------------------------------- GPU: the kernels
def _pair(ora, hip, n_extra):
    cfg = A.base_model_config(1, n_extra_dims=n_extra)
    om, hm = OraModel(ora, cfg), HipModel(hip, cfg)
    p = _fill(om)
    ora.ora_model_sync_half(om.h)
    hm.set_params(p)
    return cfg, om, hm


def _rel_l2(a, b):
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-20))


@pytest.mark.gpu
@pytest.mark.parametrize("n_extra", [3, 16])
def test_init_matches_the_oracle(ora, hip, n_extra):
    import torch
    cfg = A.base_model_config(1, n_extra_dims=n_extra)
    om, hm = OraModel(ora, cfg, seed=1337), HipModel(hip, cfg, seed=1337)
    assert om.n == hm.n and om.n_mlp == hm.n_mlp == N_BASE_MLP + 1024
    assert np.array_equal(om.params_fp, hm.read("master", torch))


@pytest.mark.gpu
@pytest.mark.parametrize("n_extra", [1, 3, 16])
def test_inference_parity(ora, hip, n_extra):
    """k_inference with the third k-step of the first colour layer vs the oracle (tolerance of test_gpu_shapes.py); the extra dims change the output"""
    import torch
    cfg, om, hm = _pair(ora, hip, n_extra)
    n = 20011
    c = _coords(n, n_extra, 13)
    ref = half_to_f32(_ora_inference(ora, om, c))
    cd = torch.from_numpy(c).cuda()
    out = torch.zeros((n, 4), dtype=torch.int16, device="cuda")
    A.check(hip, hip.ngp_model_inference(hm.h, None, dptr(cd), 7 + n_extra, n, None, dptr(out), 4, 0))
    torch.cuda.synchronize()
    got = half_to_f32(out.cpu().numpy().view(np.uint16))
    err = np.abs(got - ref)
    assert (err <= 2e-3 + 1e-2 * np.abs(ref)).all(), err.max()
    c2 = c.copy(); c2[:, 7] = -c2[:, 7]
    cd2 = torch.from_numpy(c2).cuda(); out2 = torch.zeros_like(out)
    A.check(hip, hip.ngp_model_inference(hm.h, None, dptr(cd2), 7 + n_extra, n, None, dptr(out2), 4, 0))
    torch.cuda.synchronize()
    got2 = half_to_f32(out2.cpu().numpy().view(np.uint16))
    assert np.array_equal(got[:, 3], got2[:, 3]) and np.abs(got[:, :3] - got2[:, :3]).max() > 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("n_extra", [3, 16])
def test_training_step_gradients_and_input_gradient(ora, hip, n_extra):
    """T1 / W with the wider first colour layer: parameter gradients per block vs the oracle, and dL/d(extra dims) per sample (the half matrix the colour network's
    backward produces, handed on as float by the Identity encoding) -- with the record lists and with atomics only."""
    import torch
    cfg, om, hm = _pair(ora, hip, n_extra)
    n = 1 << 15
    c = _coords(n, n_extra, 21)
    rng = np.random.default_rng(5)
    dl = (rng.normal(size=(n, 4)) * (128.0 / n)).astype(np.float16).view(np.uint16)
    dref = np.zeros((n, n_extra), np.float32)
    ora.ora_model_training_step_extra(om.h, ptr(c), 7 + n_extra, n, ptr(dl), 4, ptr(dref))
    gref = half_to_f32(om.grads.copy())
    ora.ora_model_training_step_exact_sums(om.h, ptr(c), 7 + n_extra, n, ptr(dl), 4, 2)  # the same gradients with every table entry's contributions summed without rounding
    gtrue = half_to_f32(om.grads.copy())
    cd = torch.from_numpy(c).cuda(); dld = torch.from_numpy(dl.view(np.int16)).cuda()
    blocks = {"density_l1": (0, 2048), "density_l2": (2048, 3072), "rgb_l1": (3072, 3072 + 64 * 48), "rgb_l2": (6144, 6144 + 4096), "rgb_out": (10240, 10240 + 3 * 64)}
    offs = (C.c_uint32 * 9)(); res = (C.c_uint32 * 8)(); sc = (C.c_float * 8)()
    hip.ngp_model_grid_layout(hm.h, offs, res, sc)
    levels = {l: (hm.n_mlp + offs[l] * 4, hm.n_mlp + offs[l + 1] * 4) for l in range(8)}
    noise = {l: _rel_l2(gref[a:b], gtrue[a:b]) for l, (a, b) in levels.items()}  # the reference-order chain of half adds vs the unrounded sums
    print("reference-order oracle vs unrounded sums per level", {l: f"{v:.1e}" for l, v in noise.items()})
    try:
        for vname, flags in [("lists", 0), ("atomics_only", 2048)]:
            hip.ngp_debug_set_flags(flags)
            dx = torch.full((n, n_extra), float("nan"), dtype=torch.float32, device="cuda")
            A.check(hip, hip.ngp_model_training_step_extra(hm.h, None, dptr(cd), 7 + n_extra, n, dptr(dld), 4, dptr(dx)))
            torch.cuda.synchronize()
            gf = half_to_f32(hm.read("grads", torch))
            assert np.isfinite(gf).all()
            rep = {k: _rel_l2(gf[a:b], gref[a:b]) for k, (a, b) in blocks.items()}
            w1 = gf[3072:3072 + 64 * 48].reshape(64, 48); w1r = gref[3072:3072 + 64 * 48].reshape(64, 48)
            rep["rgb_l1_extra_columns"] = _rel_l2(w1[:, 32:], w1r[:, 32:])
            got = dx.cpu().numpy()
            rep["dL_dextra"] = _rel_l2(got, dref)
            print(n_extra, vname, {k: f"{v:.1e}" for k, v in rep.items()})
            for k, v in rep.items():
                assert v < 1.5e-3, (vname, k, v)   # measured <= 6.7e-4 (GPUTEST r05 batch a)
            # the hash-grid gradient per level against the UNROUNDED sums, the plain model's bar (tests/test_gpu_model.py GRID_TRUE_TOL): the record lists sum exactly, what is
            # left is the MLP backward's half rounding of dL/d(enc) -- measured 2.8e-4 .. 1.2e-3 per level (n_extra 3 and 16; GPUTEST r05 batch a), bound 2 x that.  (At 2^15
            # samples few entries collide, so the reference-order chain of half adds is nearly exact as well -- `noise` 2.4e-4 .. 2.5e-3 -- and the plain test's "at least as
            # close as the reference order" statement, made at 2^18 samples, has nothing to separate here.)  Half atomics carry the reference's own kind of error on top.
            true = {l: _rel_l2(gf[a:b], gtrue[a:b]) for l, (a, b) in levels.items()}
            print(n_extra, vname, "grid vs unrounded sums per level", {l: f"{v:.1e}" for l, v in true.items()})
            for l, v in true.items():
                if vname == "lists":
                    assert v < 2.5e-3, (vname, l, v)
                else:
                    assert v < 2.0 * noise[l] + 2.5e-3, (vname, l, v, noise[l])
            assert np.isfinite(got).all() and np.abs(w1r[:, 32:]).max() > 0
            # per element: both sides round the same fp32 sums of 64 products to half
            assert (np.abs(got - dref) <= 2e-3 * np.abs(dref) + 1e-7).mean() > 0.99
    finally:
        hip.ngp_debug_set_flags(0)


def _tinted_dataset(n_img, res, n_poses=None):
    """the small synthetic scene with a per-image colour cast that only a per-image input can explain.  n_poses < n_img: image i is taken from pose i % n_poses, so
    several images share one camera and differ by their cast alone (a view-dependent colour cannot tell them apart, an appearance vector can)"""
    n_poses = n_poses or n_img
    imgs0, xforms0, meta = make_small_dataset(n_poses, res)
    imgs = [imgs0[i % n_poses] for i in range(n_img)]
    xforms = [xforms0[i % n_poses] for i in range(n_img)]
    rng = np.random.default_rng(11)
    tints = rng.uniform(0.55, 1.0, (n_img, 3)).astype(np.float32)
    out = []
    for im, t in zip(imgs, tints):
        a = im.reshape(res, res, 4).astype(np.float32)
        a[..., :3] *= t
        out.append(np.ascontiguousarray(a.astype(np.uint8).reshape(im.shape)))
    return out, xforms, meta, tints


def _trainer(hip, cfg, imgs, xforms, meta, batch=1 << 16):
    M, X = host_meta(imgs, xforms, meta)
    hm = HipModel(hip, cfg)
    t = C.c_void_p()
    opts = A.default_nerf_options(1, target_batch_size=batch)
    A.check(hip, hip.ngp_nerf_create(hm.h, C.byref(opts), A.scene_aabb(1), C.byref(t)))
    pix = (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
    A.check(hip, hip.ngp_nerf_set_dataset_host(t, len(imgs), M, X, pix))
    return hm, t, (M, X, pix)


def _loss(hip, t):
    st = A.NerfStats(); A.check(hip, hip.ngp_nerf_get_stats(t, None, C.byref(st)))
    return float(st.loss), st


@pytest.mark.gpu
def test_trainer_with_extra_dims(hip):
    """The NeRF trainer with a model that has extra dims: K1 copies every image's vector behind its rays' samples, the step runs (eager K2, generic T1 / W), the per-image
    gradient reaches the latents when optimize_extra_dims is on and nothing moves when it is off; fixed light-direction dims (has_light_dirs) are inputs only."""
    n_img, n_extra = 12, 4
    imgs, xforms, meta, tints = _tinted_dataset(n_img, 96)
    cfg = A.base_model_config(1, n_extra_dims=n_extra)
    hm, t, keep = _trainer(hip, cfg, imgs, xforms, meta)
    rng = np.random.default_rng(2)
    e0 = rng.uniform(-1, 1, (n_img, n_extra)).astype(np.float32)  # reset_extra_dims: random_val(rng) * 2 - 1
    A.check(hip, hip.ngp_nerf_set_extra_dims(t, ptr(e0), n_img))
    back = np.zeros_like(e0); A.check(hip, hip.ngp_nerf_get_extra_dims(t, ptr(back), n_img))
    assert np.array_equal(back, e0)
    # off: the latents are inputs only
    A.check(hip, hip.ngp_nerf_train(t, None, 20))
    l_first, st = _loss(hip, t)
    A.check(hip, hip.ngp_nerf_get_extra_dims(t, ptr(back), n_img))
    assert np.array_equal(back, e0) and np.isfinite(l_first) and st.measured_batch_size > 0
    # on: gradient per image, Adam moves every image's vector
    A.check(hip, hip.ngp_nerf_set_optimize_extra_dims(t, 1))
    A.check(hip, hip.ngp_nerf_train(t, None, 1))
    g = np.zeros_like(e0); A.check(hip, hip.ngp_nerf_get_extra_dims_gradient(t, ptr(g), n_img))
    e1 = np.zeros_like(e0); A.check(hip, hip.ngp_nerf_get_extra_dims(t, ptr(e1), n_img))
    assert np.isfinite(g).all() and (np.abs(g).max(axis=1) > 0).all(), "every image has rays in a batch, so every image has a gradient"
    lr = float(hip.ngp_model_learning_rate(hm.h))
    # first VarAdam step from zero moments (adam_optimizer.h:37-47): m = 0.1 g, sqrt(v) = 0.1 |g|, debiasing factor sqrt(1 - 0.99) / (1 - 0.9) = 1 => delta = lr * 0.1 g / (0.1 |g| + 1e-8)
    gs = g.astype(np.float64) / 128.0
    assert np.allclose(e0 - e1, lr * 0.1 * gs / (0.1 * np.abs(gs) + 1e-8), rtol=1e-3, atol=1e-7)
    A.check(hip, hip.ngp_nerf_train(t, None, 150))
    l_end, _ = _loss(hip, t)
    e2 = np.zeros_like(e0); A.check(hip, hip.ngp_nerf_get_extra_dims(t, ptr(e2), n_img))
    print(f"loss {l_first:.5f} -> {l_end:.5f}; latent drift {np.abs(e2 - e0).mean():.4f}")
    assert np.isfinite(e2).all() and l_end < 0.5 * l_first
    hip.ngp_nerf_destroy(t)


@pytest.mark.gpu
def test_latents_explain_a_per_image_colour_cast(hip):
    """What the extra dims are for (appearance embeddings): on images with a per-image colour cast a model with 4 learnable dims per image reaches a clearly lower training
    loss than base.json's plain model after the same number of steps.  Pairs of images share one camera pose: the plain model's best answer for such a pair is the mean of
    the two casts (a loss floor of about (c dt / 2)^2), the latents are the only input that separates them."""
    n_img = 12
    imgs, xforms, meta, tints = _tinted_dataset(n_img, 96, n_poses=6)
    losses = {}
    for name, n_extra in (("plain", 0), ("latents", 4)):
        cfg = A.base_model_config(1, n_extra_dims=n_extra)
        hm, t, keep = _trainer(hip, cfg, imgs, xforms, meta)
        if n_extra:
            e0 = np.random.default_rng(2).uniform(-1, 1, (n_img, n_extra)).astype(np.float32)
            A.check(hip, hip.ngp_nerf_set_extra_dims(t, ptr(e0), n_img))
            A.check(hip, hip.ngp_nerf_set_optimize_extra_dims(t, 1))
        A.check(hip, hip.ngp_nerf_train(t, None, 600))
        ls = []
        for _ in range(8):
            A.check(hip, hip.ngp_nerf_train(t, None, 5)); ls.append(_loss(hip, t)[0])
        losses[name] = float(np.mean(ls))
        hip.ngp_nerf_destroy(t)
    print(losses)
    assert losses["latents"] < 0.5 * losses["plain"], losses


@pytest.mark.gpu
def test_rendering_extra_dims(hip):
    """Testbed::Nerf::get_rendering_extra_dims (testbed_nerf.cu:3685-3707): a rendering uses one vector for the whole frame -- a training view's, or explicit values;
    the same vector through both routes gives the same frame, another view's a different one."""
    import torch
    n_img, n_extra = 12, 4
    imgs, xforms, meta, tints = _tinted_dataset(n_img, 96)
    cfg = A.base_model_config(1, n_extra_dims=n_extra)
    hm, t, keep = _trainer(hip, cfg, imgs, xforms, meta)
    e0 = np.random.default_rng(2).uniform(-1, 1, (n_img, n_extra)).astype(np.float32)
    A.check(hip, hip.ngp_nerf_set_extra_dims(t, ptr(e0), n_img))
    A.check(hip, hip.ngp_nerf_set_optimize_extra_dims(t, 1))
    A.check(hip, hip.ngp_nerf_train(t, None, 300))
    e = np.zeros_like(e0); A.check(hip, hip.ngp_nerf_get_extra_dims(t, ptr(e), n_img))
    res = 64
    M, X, _ = keep
    rp = A.RenderParams()
    rp.resolution[0] = rp.resolution[1] = res
    rp.focal_length[0] = rp.focal_length[1] = M[0].focal_length[0] * res / M[0].resolution[0]
    rp.screen_center[0] = rp.screen_center[1] = 0.5
    for k in range(12):
        rp.camera[k] = X[0].start[k]
    rp.lens_mode = 0; rp.spp_index = 0; rp.snap_to_pixel_centers = 1; rp.min_transmittance = 1e-4; rp.near_distance = 0.0; rp.use_inference_params = 0
    rp.render_aabb = A.scene_aabb(1)
    frames = {}
    for name, view, vals in (("view0", 0, None), ("explicit0", -1, e[0]), ("view5", 5, None)):
        A.check(hip, hip.ngp_nerf_set_rendering_extra_dims(t, view, ptr(np.ascontiguousarray(vals)) if vals is not None else None))
        fb = torch.zeros((res * res, 4), dtype=torch.float32, device="cuda"); db = torch.zeros(res * res, dtype=torch.float32, device="cuda")
        A.check(hip, hip.ngp_nerf_render(t, None, C.byref(rp), dptr(fb), dptr(db)))
        torch.cuda.synchronize()
        frames[name] = fb.cpu().numpy()
    assert np.isfinite(frames["view0"]).all() and frames["view0"][:, 3].max() > 0.5
    assert np.array_equal(frames["view0"], frames["explicit0"])
    assert np.abs(frames["view0"][:, :3] - frames["view5"][:, :3]).max() > 1e-3
    hip.ngp_nerf_destroy(t)
