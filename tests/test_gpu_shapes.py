"""GPU parity of the fused hash-grid + MLP kernels for the network shapes beyond configs/nerf/base.json (reference: reset_network builds the model
from the json, testbed.cu:4160-4412; shipped configs/nerf/{base_14, small, big, hashgrid}.json and the 2022 base.json = HashGrid L = 16, F = 2
that notebooks/instant_ngp.ipynb:5838 ran): encoding bit-exact, outputs / gradients vs the oracle with the tolerances of test_gpu_model.py, every
scatter layout of the record lists, and a short training run per shape."""
import ctypes as C

import numpy as np
import pytest

import ngp_abi as A
from common import HipModel, OraModel, dptr, half_to_f32, host_meta, make_small_dataset, ptr, random_coords

pytestmark = pytest.mark.gpu

# name -> (aabb_scale, ModelConfig overrides)
SHAPES = {
    "l16f2_t19": (1, dict(n_levels=16, n_features_per_level=2)),                          # notebook configuration at aabb_scale 1
    "l16f2_t19_aabb4": (4, dict(n_levels=16, n_features_per_level=2)),                    # ... and at fox's aabb_scale: b = 1.51572
    "l8f4_t14": (1, dict(log2_hashmap_size=14)),                                          # configs/nerf/base_14.json
    "l8f4_t15": (1, dict(log2_hashmap_size=15)),                                          # configs/nerf/small.json
    "l8f4_t21": (1, dict(log2_hashmap_size=21)),                                          # configs/nerf/big.json
    "l16f2_t15": (1, dict(n_levels=16, n_features_per_level=2, log2_hashmap_size=15)),
    "l8f4_rgb1": (1, dict(n_hidden_layers_rgb=1)),                                        # configs/nerf/base_1layer.json
    "l8f4_rgb3": (1, dict(n_hidden_layers_rgb=3)),                                        # configs/nerf/base_3layer.json
    "l16f2_rgb3_t15": (1, dict(n_levels=16, n_features_per_level=2, log2_hashmap_size=15, n_hidden_layers_rgb=3)),
}


def _mlp_blocks(cfg):
    """parameter blocks of the two MLPs in tcnn order (nerf_network.h:357-372) and the total"""
    nr = cfg.n_hidden_layers_rgb
    b = {"density_l1": (0, 2048), "density_l2": (2048, 3072), "rgb_l1": (3072, 5120)}
    for k in range(nr - 1):
        b[f"rgb_l2_{k}"] = (5120 + 4096 * k, 5120 + 4096 * (k + 1))
    o = 5120 + 4096 * (nr - 1)
    b["rgb_out"] = (o, o + 3 * 64)  # rows 3..15 of the padded output layer receive no gradient
    return b, o + 1024


def _cfg(name):
    aabb, kw = SHAPES[name]
    return A.base_model_config(aabb, **kw)


def _models(ora, hip, name):
    cfg = _cfg(name)
    om = OraModel(ora, cfg)
    hm = HipModel(hip, cfg)
    rng = np.random.default_rng(7)
    p = om.params_fp
    p[: om.n_mlp] = rng.uniform(-0.3, 0.3, om.n_mlp).astype(np.float32)
    p[om.n_mlp:] = rng.uniform(-1.0, 1.0, om.n - om.n_mlp).astype(np.float32)
    ora.ora_model_sync_half(om.h)
    hm.set_params(p)
    return cfg, om, hm


def _layout(hip, hm, cfg):
    L, F = cfg.n_levels, cfg.n_features_per_level
    offs = (C.c_uint32 * (L + 1))(); res = (C.c_uint32 * L)(); sc = (C.c_float * L)()
    hip.ngp_model_grid_layout(hm.h, offs, res, sc)
    return L, F, [int(o) for o in offs], [int(r) for r in res], [float(s) for s in sc]


def _rel_l2(a, b):
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-20))


@pytest.mark.parametrize("name", list(SHAPES))
def test_init_and_layout(ora, hip, name):
    import torch
    cfg = _cfg(name)
    om = OraModel(ora, cfg, seed=1337)
    hm = HipModel(hip, cfg, seed=1337)
    n_mlp = _mlp_blocks(cfg)[1]
    assert om.n == hm.n and om.n_mlp == hm.n_mlp == n_mlp
    assert np.array_equal(om.params_fp, hm.read("master", torch))
    L, F, offs, res, sc = _layout(hip, hm, cfg)
    assert hm.n == n_mlp + offs[L] * F
    if name == "l16f2_t19_aabb4":  # the reference's own log: `GridEncoding: Nmin=16 b=1.51572 F=2 T=2^19 L=16`, total_encoding_params=13074912
        assert offs[L] * F == 13074912 and abs(cfg.per_level_scale - 1.51572) < 5e-6


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("n,coherent", [(31, False), (4096, False), (50003, True)])
def test_encode_bit_exact(ora, hip, name, n, coherent):
    import torch
    cfg, om, hm = _models(ora, hip, name)
    c = random_coords(n, seed=n, ray_coherent=coherent)
    c[0, 0:3] = (1.0, 1.0, 1.0)  # upper boundary: the dense-level index wrap
    ref = om.encode(c)
    cd = torch.from_numpy(c).cuda()
    out = torch.zeros((n, 32), dtype=torch.int16, device="cuda")
    A.check(hip, hip.ngp_model_encode(hm.h, None, dptr(cd), 7, n, dptr(out)))
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint16)
    bad = np.argwhere(ref != got)
    assert len(bad) == 0, f"{len(bad)} of {ref.size} encoded halfs differ, first {bad[:4].tolist()}"


@pytest.mark.parametrize("name", list(SHAPES))
def test_inference_and_density_parity(ora, hip, name):
    import torch
    cfg, om, hm = _models(ora, hip, name)
    n = 20011
    c = random_coords(n, seed=13, ray_coherent=True)
    ref = half_to_f32(om.inference(c))
    cd = torch.from_numpy(c).cuda()
    out = torch.zeros((n, 4), dtype=torch.int16, device="cuda")
    A.check(hip, hip.ngp_model_inference(hm.h, None, dptr(cd), 7, n, None, dptr(out), 4, 0))
    torch.cuda.synchronize()
    got = half_to_f32(out.cpu().numpy().view(np.uint16))
    err = np.abs(got - ref)
    assert (err <= 2e-3 + 1e-2 * np.abs(ref)).all(), err.max()
    pos = np.ascontiguousarray(c[:, :3])
    dref = half_to_f32(om.density(pos))
    pd = torch.from_numpy(pos).cuda()
    dout = torch.zeros((n,), dtype=torch.int16, device="cuda")
    A.check(hip, hip.ngp_model_density(hm.h, None, dptr(pd), 3, n, dptr(dout), 1, 0))
    torch.cuda.synchronize()
    dgot = half_to_f32(dout.cpu().numpy().view(np.uint16))
    assert (np.abs(dgot - dref) <= 2e-3 + 1e-2 * np.abs(dref)).all()


@pytest.mark.parametrize("name", list(SHAPES))
def test_training_step_gradients_all_scatter_layouts(ora, hip, name):
    """HIP vs oracle per parameter block at 2^16 ray-coherent samples: the production layout (record lists where the table sizes allow them), the
    2^11-entry chunks, a forced list overflow and the atomics-only path.  The list layouts sum exactly => bit-identical among themselves."""
    import torch
    cfg, om, hm = _models(ora, hip, name)
    n = 1 << 16
    c = random_coords(n, seed=21, ray_coherent=True)
    rng = np.random.default_rng(5)
    dl = (rng.normal(size=(n, 4)) * (128.0 / n)).astype(np.float16).view(np.uint16)
    om.training_step(c, dl)
    gref = half_to_f32(om.grads.copy())
    cd = torch.from_numpy(c).cuda(); dld = torch.from_numpy(dl.view(np.int16)).cuda()
    L, F, offs, res, sc = _layout(hip, hm, cfg)
    blocks, n_mlp = _mlp_blocks(cfg)
    for l in range(L):
        blocks[f"grid_level_{l}"] = (n_mlp + offs[l] * F, n_mlp + offs[l + 1] * F)
    got = {}
    try:
        for vname, cl2, cap, flags in [("chunk12", 12, 0, 0), ("chunk11", 11, 0, 0), ("chunk12_overflow", 12, 2048, 0), ("atomics_only", 12, 0, 2048), ("w_single_role", 12, 0, 32768)]:
            A.check(hip, hip.ngp_debug_set_bin_params(cl2, 0, cap)); hip.ngp_debug_set_flags(flags)
            A.check(hip, hip.ngp_model_training_step(hm.h, None, dptr(cd), 7, n, dptr(dld), 4))
            torch.cuda.synchronize()
            g = hm.read("grads", torch).copy()
            got[vname] = g
            gf = half_to_f32(g)
            assert np.isfinite(gf).all()
            report = {k: _rel_l2(gf[a:b], gref[a:b]) for k, (a, b) in blocks.items() if np.linalg.norm(gref[a:b]) > 0}
            print(name, vname, {k: f"{v:.1e}" for k, v in report.items()})
            for k, v in report.items():
                assert v < (2e-2 if not k.startswith("grid") else 5e-2), (name, vname, k, v)
        assert np.all(half_to_f32(got["chunk12"])[blocks["rgb_out"][1]:n_mlp] == 0)  # output-layer rows 3..15 receive no gradient
        big = np.abs(gref[n_mlp:]) > 1e-4
        assert np.all(got["chunk12"][n_mlp:][big] != 0)
        # the hashed levels of every list layout hold the exact sum of the same records
        for l in range(L):
            lo, hi_ = blocks[f"grid_level_{l}"]
            if res[l] ** 3 > offs[l + 1] - offs[l] and (1 << 12) <= offs[l + 1] - offs[l] <= (1 << 19):  # hashed and within the lists' table sizes
                assert np.array_equal(got["chunk12"][lo:hi_], got["chunk11"][lo:hi_]), (name, "chunk11 != chunk12", l)
    finally:
        A.check(hip, hip.ngp_debug_set_bin_params(11, 0, 0)); hip.ngp_debug_set_flags(0)  # (11: the library default since round 6)


@pytest.mark.parametrize("name", ["l16f2_t19", "l8f4_t15", "l8f4_t21", "l16f2_t15", "l8f4_rgb1", "l8f4_rgb3", "l16f2_rgb3_t15"])
def test_short_training_run(hip, name):
    """each shape creates, trains and learns: the per-batch loss falls below HALF of its initial value within 150 steps of the small synthetic scene.  (Measured over the
    rounds' tiers: the ratio is 3.7 - 5.0 for every shape in nearly every run, but training is not bit-reproducible -- K3 reserves its compacted spans by atomics -- and one
    run of `l16f2_rgb3_t15` in round 5 ended at 2.99 against the former bar of 3, with the same library that had given 3.7 twenty minutes earlier.)"""
    import torch
    imgs, xforms, meta = make_small_dataset(12, 96)
    M, X = host_meta(imgs, xforms, meta)
    aabb, kw = SHAPES[name]
    cfg = _cfg(name)
    hm = HipModel(hip, cfg)
    t = C.c_void_p()
    opts = A.default_nerf_options(1, target_batch_size=1 << 16)
    A.check(hip, hip.ngp_nerf_create(hm.h, C.byref(opts), A.scene_aabb(1), C.byref(t)))
    pix = (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
    A.check(hip, hip.ngp_nerf_set_dataset_host(t, len(imgs), M, X, pix))
    # deterministic compaction (DBG_K3_TWO_PASS: the production K3 orders the batch rows by atomic arrival, so two runs differ in the last bits and this loss ratio
    # scattered around its bar -- a flake of the round-5 final tier): with slot-ordered rows the run is the same every time and the round-3 bar (a factor 3 in 150 steps) is back
    hip.ngp_debug_set_flags(1048576)
    try:
        losses = []
        for _ in range(6):
            A.check(hip, hip.ngp_nerf_train(t, None, 25))
            st = A.NerfStats(); A.check(hip, hip.ngp_nerf_get_stats(t, None, C.byref(st)))
            losses.append(float(st.loss))
    finally:
        hip.ngp_debug_set_flags(0)
    print(name, [f"{l:.5f}" for l in losses])
    assert np.isfinite(losses).all() and losses[-1] < losses[0] / 3 and st.measured_batch_size > 0
    hip.ngp_nerf_destroy(t)
