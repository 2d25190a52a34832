#!/usr/bin/env python
"""What the Appendix-A switch `mlp_half_accumulate` is worth: tcnn's FullyFusedMLP may keep its WMMA accumulator fragments in __half (one rounding
per 16-wide k-step) where this repository's MFMA kernels -- and the oracle by default -- accumulate the whole contraction in fp32 (one rounding per output).  tcnn's source is
absent from the mount, so the question cannot be settled; this tool puts numbers on the difference with the ORACLE (CPU) on a state trained by the product path (GPU):

  1. the product trains the lego-format stand-in (B = 2^16, --steps steps); the trained parameters and occupancy grid go into two oracle trainers;
  2. RENDER A/B: the oracle renders held-out views with fp32 and with half accumulators from the SAME weights -> PSNR each, and the PSNR of one rendering against the other;
  3. OUTPUT / GRADIENT A/B: network outputs on 2^15 ray-coherent samples and one training step's gradients, half vs fp32 accumulators (relative L2 per parameter block);
  4. TRAINING A/B: --cpu-steps further steps at B = 2^14 on the CPU in each mode from the same state (same rays), then the same render -> PSNR each.

usage (GPU box): python tools/ab_half_accumulate.py [--steps 1500] [--cpu-steps 24] [--res 96] > out.json"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "instant-ngp_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ngp_abi as A  # noqa: E402
import oracle_py  # noqa: E402
import synth_scene  # noqa: E402
from common import OraModel, ptr, random_coords, half_to_f32  # noqa: E402


def psnr(a, b):
    m = float(((a - b) ** 2).mean())
    return -10.0 * math.log10(m) if m > 0 else float("inf")


def srgb(lin):
    lin = np.clip(lin, 0, 1)
    return np.clip(np.where(lin < 0.0031308, 12.92 * lin, 1.055 * np.maximum(lin, 1e-12) ** (1 / 2.4) - 0.055), 0, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--cpu-steps", type=int, default=24)
    ap.add_argument("--res", type=int, default=96)
    ap.add_argument("--views", type=int, default=3)
    ap.add_argument("--images", type=int, default=40)
    ap.add_argument("--train-res", type=int, default=200)
    args = ap.parse_args()
    lib = A.load_hip(); ora = oracle_py.load()
    ora.ora_set_mlp_half_accumulate.argtypes = [C.c_int]
    A.check(lib, lib.ngp_init())
    images, xforms, meta, _ = synth_scene.make_dataset(args.images, args.train_res, "cuda")
    n = len(images)
    M = (A.ImageMeta * n)(); X = (A.Xform * n)()
    for i in range(n):
        M[i].pixels = images[i].data_ptr(); M[i].image_data_type = A.IMAGE_BYTE; M[i].lens_mode = 0
        M[i].resolution[0] = M[i].resolution[1] = args.train_res; M[i].principal_point[0] = M[i].principal_point[1] = 0.5
        M[i].focal_length[0], M[i].focal_length[1] = meta["focal_length"]
        for k in range(12):
            X[i].start[k] = X[i].end[k] = float(xforms[i][k])
    cfg = A.base_model_config(1)
    model = C.c_void_p(); A.check(lib, lib.ngp_model_create(C.byref(cfg), C.c_uint64(1337), C.byref(model)))
    opts = A.default_nerf_options(1, target_batch_size=1 << 16)
    nerf = C.c_void_p(); A.check(lib, lib.ngp_nerf_create(model, C.byref(opts), A.scene_aabb(1), C.byref(nerf)))
    A.check(lib, lib.ngp_nerf_set_dataset_device(nerf, n, M, X))
    A.check(lib, lib.ngp_nerf_train(nerf, None, args.steps)); torch.cuda.synchronize()
    st = A.NerfStats(); A.check(lib, lib.ngp_nerf_get_stats(nerf, None, C.byref(st)))
    n_params = C.c_uint64(); lib.ngp_model_n_params(model, C.byref(n_params), None)
    params = np.empty(n_params.value, np.float32)
    A.check(lib, lib.ngp_model_get_params_host(model, ptr(params), C.c_uint64(params.size)))
    gp = C.c_void_p(); lib.ngp_nerf_density_grid_ptrs(nerf, C.byref(gp), None, None)
    grid = np.empty(128 ** 3, np.float32)
    rt = C.CDLL("libamdhip64.so"); assert rt.hipMemcpy(ptr(grid), gp, C.c_size_t(grid.nbytes), 2) == 0
    host_imgs = [im.cpu().numpy() for im in images]
    Mh = (A.ImageMeta * n)()
    for i in range(n):
        C.memmove(C.byref(Mh[i]), C.byref(M[i]), C.sizeof(A.ImageMeta)); Mh[i].pixels = host_imgs[i].ctypes.data
    o2 = A.default_nerf_options(1, target_batch_size=1 << 14)

    def oracle_trainer():
        om = OraModel(ora, cfg)
        om.params_fp[:] = params; ora.ora_model_sync_half(om.h)
        ot = C.c_void_p(); assert ora.ora_nerf_create(om.h, C.byref(o2), A.scene_aabb(1), C.byref(ot)) == 0
        ora.ora_nerf_set_dataset(ot, n, Mh, X)
        C.memmove(ora.ora_nerf_density_grid(ot), grid.ctypes.data, grid.nbytes)
        ora.ora_nerf_update_mean_and_bitfield(ot)
        ora.ora_nerf_set_rays_per_batch(ot, max(256, int(st.rays_per_batch * (1 << 14) / (1 << 16)) // 256 * 256))
        ora.ora_nerf_set_training_step(ot, st.training_step)
        return om, ot

    gts, exf, emeta, _ = synth_scene.make_dataset(args.views, args.res, "cpu", phase=1.234)

    def render_all(ot):
        out = []
        for gt, xf in zip(gts, exf):
            rp = A.RenderParams()
            rp.resolution[0] = rp.resolution[1] = args.res
            rp.focal_length[0], rp.focal_length[1] = emeta["focal_length"]; rp.screen_center[0] = rp.screen_center[1] = 0.5
            for k in range(12):
                rp.camera[k] = float(xf[k])
            rp.lens_mode = 0; rp.spp_index = 0; rp.snap_to_pixel_centers = 1; rp.min_transmittance = 1e-4; rp.near_distance = 0.0; rp.use_inference_params = 0
            rp.render_aabb = A.scene_aabb(1)
            f = np.zeros((args.res * args.res, 4), np.float32); dep = np.zeros(args.res * args.res, np.float32)
            assert ora.ora_nerf_render(ot, C.byref(rp), ptr(f), ptr(dep)) == 0
            out.append(srgb(f[:, :3]))
        return out

    def gt_srgb():
        out = []
        for gt in gts:
            g = gt.numpy().reshape(-1, 4).astype(np.float32) / 255.0
            lin = np.where(g[:, :3] <= 0.04045, g[:, :3] / 12.92, ((g[:, :3] + 0.055) / 1.055) ** 2.4) * g[:, 3:4]
            out.append(srgb(lin))
        return out
    G = gt_srgb()
    res = {"what": "oracle with fp32 (default) vs half (tcnn WMMA __half fragments, modelled) matrix-multiply accumulators, from one product-trained state",
           "product_training": {"steps": args.steps, "batch": 1 << 16, "scene": f"lego-format stand-in, {n} views {args.train_res}^2", "loss": st.loss},
           "render": {"views": args.views, "resolution": args.res, "kind": "held-out synthetic test cameras, training (not EMA) weights, spp 1"}}
    t0 = time.time()
    om0, ot0 = oracle_trainer(); om1, ot1 = oracle_trainer()
    ora.ora_set_mlp_half_accumulate(0); R0 = render_all(ot0)
    ora.ora_set_mlp_half_accumulate(1); R1 = render_all(ot1)
    res["render_ab"] = {"psnr_fp32_acc_db": round(float(np.mean([psnr(a, g) for a, g in zip(R0, G)])), 4), "psnr_half_acc_db": round(float(np.mean([psnr(a, g) for a, g in zip(R1, G)])), 4),
                        "psnr_half_vs_fp32_rendering_db": round(float(np.mean([psnr(a, b) for a, b in zip(R0, R1)])), 2), "seconds": round(time.time() - t0, 1)}
    res["render_ab"]["delta_db"] = round(res["render_ab"]["psnr_half_acc_db"] - res["render_ab"]["psnr_fp32_acc_db"], 4)
    # outputs and gradients on the same inputs
    c = random_coords(1 << 15, seed=5, ray_coherent=True)
    ora.ora_set_mlp_half_accumulate(0); o0 = half_to_f32(om0.inference(c))
    ora.ora_set_mlp_half_accumulate(1); o1 = half_to_f32(om1.inference(c))
    rng = np.random.default_rng(3)
    dl = (rng.normal(size=(1 << 15, 4)) * (128.0 / (1 << 15))).astype(np.float16).view(np.uint16)
    ora.ora_set_mlp_half_accumulate(0); om0.training_step(c, dl); g0 = half_to_f32(om0.grads.copy())
    ora.ora_set_mlp_half_accumulate(1); om1.training_step(c, dl); g1 = half_to_f32(om1.grads.copy())

    def rel(a, b):
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-30))
    nm = om0.n_mlp
    res["numeric_ab"] = {"outputs_rel_l2": rel(o0, o1), "outputs_max_abs": float(np.abs(o0 - o1).max()), "mlp_gradient_rel_l2": rel(g0[:nm], g1[:nm]), "grid_gradient_rel_l2": rel(g0[nm:], g1[nm:]),
                         "note": "for scale: the product's outputs differ from the fp32-accumulator oracle by <= 2e-3 + 1 % (tests/test_gpu_model.py)"}
    # training A/B on the CPU from the same state, same ray stream
    if args.cpu_steps > 0:
        t0 = time.time()
        losses = {}
        for mode, ot in ((0, ot0), (1, ot1)):
            ora.ora_set_mlp_half_accumulate(mode)
            ls = []
            for _ in range(args.cpu_steps):
                assert ora.ora_nerf_train_forward_backward(ot) == 0 and ora.ora_nerf_train_finish(ot) == 0, ora.ora_last_error()
                s = A.NerfStats(); ora.ora_nerf_get_stats(ot, C.byref(s)); ls.append(float(s.loss))
            losses[mode] = ls
        ora.ora_set_mlp_half_accumulate(0); T0 = render_all(ot0)
        ora.ora_set_mlp_half_accumulate(1); T1 = render_all(ot1)
        res["training_ab"] = {"cpu_steps": args.cpu_steps, "batch": 1 << 14, "mean_loss_fp32_acc": float(np.mean(losses[0])), "mean_loss_half_acc": float(np.mean(losses[1])),
                              "psnr_fp32_acc_db": round(float(np.mean([psnr(a, g) for a, g in zip(T0, G)])), 4), "psnr_half_acc_db": round(float(np.mean([psnr(a, g) for a, g in zip(T1, G)])), 4),
                              "seconds": round(time.time() - t0, 1)}
        res["training_ab"]["delta_db"] = round(res["training_ab"]["psnr_half_acc_db"] - res["training_ab"]["psnr_fp32_acc_db"], 4)
    ora.ora_set_mlp_half_accumulate(0)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
