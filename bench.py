#!/usr/bin/env python
"""bench.py -- NeRF training throughput of the MI355X hot path (BASELINE.json: training rays/sec on lego, B = 2^18).

`python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by torch.distributed.run (one rank per GPU, RCCL).
A "step" = one Testbed::train(batch) call: occupancy-grid prep at the reference's cadence + K1..K6 (full optimizer step).
Data: synthetic stand-in for nerf_synthetic/lego (100 views 800x800 RGBA8, same cameras/format; the real set is not
shipped and there is no network), or `--scene fox` = the reference's shipped real capture (50 JPEGs 1080x1920, staged under
_ref_data/ by tools/stage_reference_data.py).  Before the W warm-up steps the model is trained for --pretrain steps (untimed
setup), because rays/step adapts to the occupancy grid: at initialisation one step marches ~500 rays, in steady state tens of
thousands -- the steady state is the regime the metric is quoted on.  Prints ONE JSON line on rank 0.

Untimed legs after the timed region: per-kernel HIP-event profile (roofline), held-out PSNR (scripts/run.py procedure),
`--psnr-steps` PSNR@step curve, `--ab-psnr` equal-step PSNR of the production path vs the reference-order path
(sequential K1 / K3, half atomics, eager K2: ngp_debug_set_flags(1|32|2048|8192)), CPU baseline (the oracle, a port).
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "instant-ngp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np
import torch

import ngp_abi as A

# algorithmic work per unit, SURVEY.md 8(d) / DESIGN.md "roofline accounting"
BYTES_PER_SAMPLE_FWD = 28 + 512 + 8          # coords + 8 levels x 8 corners x 8 B gather + rgbsigma half4
BYTES_PER_SAMPLE_T1_GATHER = 28 + 512 + 8 + 1024  # T1 that gathers its own encodings (every configuration but base.json's shape): forward gather + dL/dy + scatter as read-modify-write
BYTES_PER_SAMPLE_T1 = 28 + 64 + 8 + 1024     # production T1 (base.json): reads the 64-byte encoding K2 left behind instead of gathering 512 B (those gathers are K2's and charged there)
BYTES_PER_SAMPLE_K3 = 38                     # per NETWORK EVALUATION: rgbsigma half4 in + (compacted) coords 28 + dL/dy 8 out, amortised; K3 + K4 are VALU bound, listed for the whole-step sum
BYTES_PER_PARAM_OPT = 38
FLOP_PER_SAMPLE_FWD = 2 * (3072 + 7168)      # both MLPs, one sample, forward
FLOP_PER_SAMPLE_TRAIN = 3 * FLOP_PER_SAMPLE_FWD  # forward + dgrad + wgrad
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec peak)
MFMA_PEAK_TFLOPS = 2500.0                    # dense fp16 MFMA peak (no sparsity)
REFERENCE_ORDER_FLAGS = 1 | 32 | 2048 | 8192  # sequential K1, sequential K3, half atomics for every level, eager K2
CLAMP_MIN_MAX_FLAG = 2147483648  # DBG_K1_MIP_CLAMP_MIN_MAX (csrc/ngp_kernels.hpp)


class CudaView:
    """zero-copy torch view of library-owned device memory (for the all-reduce of gradients / counters through torch.distributed)"""
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


# ------------------------------------------------------------------------------------------------------------------------
# scenes
# ------------------------------------------------------------------------------------------------------------------------
LEGO_CANDIDATES = ("data/nerf/nerf_synthetic/lego", "_ref_data/data/nerf/nerf_synthetic/lego", "data/nerf_synthetic/lego", "_ref_data/data/nerf_synthetic/lego")


def find_lego():
    """scripts/scenes.py:25-32, 53: the reference keeps nerf_synthetic under data/nerf/nerf_synthetic/<scene>/transforms_train.json.  The set is not shipped with the reference and there
    is no network, so this normally returns None and the stand-in is used; NGP_LEGO_DIR or a copy under data/ (or the staged _ref_data/) makes the headline run on the real set."""
    cands = ([os.environ["NGP_LEGO_DIR"]] if os.environ.get("NGP_LEGO_DIR") else []) + [os.path.join(ROOT, c) for c in LEGO_CANDIDATES]
    for d in cands:
        f = os.path.join(d, "transforms_train.json")
        if os.path.exists(f):
            return f
    return None


def _scene_from_testbed_dataset(t, args, name, test_transforms=None):
    """a dataset loaded by this repo's C++ loader (host/testbed.cpp load_training_data) -> the device-resident scene dict"""
    import pyngp as ngp
    d = t.nerf.training.dataset
    n = d.n_images
    images = [torch.from_numpy(np.ascontiguousarray(d.image(i))).cuda() for i in range(n)]
    M = (A.ImageMeta * n)(); X = (A.Xform * n)()
    for i in range(n):
        m = d.metadata[i]
        M[i].pixels = images[i].data_ptr(); M[i].image_data_type = A.IMAGE_BYTE; M[i].lens_mode = m.lens_mode
        M[i].resolution[0], M[i].resolution[1] = m.resolution
        M[i].principal_point[0], M[i].principal_point[1] = m.principal_point
        M[i].focal_length[0], M[i].focal_length[1] = m.focal_length
        for k in range(7):
            M[i].lens_params[k] = m.lens_params[k]
        for k in range(12):
            X[i].start[k] = X[i].end[k] = d.xforms[i][k]

    def view(img, Mi, Xi):
        # Testbed::set_camera_to_training_view (testbed.cu:486-505): camera = xform, screen_center = 1 - principal point, lens of the view
        rp = A.RenderParams()
        rp.resolution[0], rp.resolution[1] = Mi.resolution[0], Mi.resolution[1]
        rp.focal_length[0], rp.focal_length[1] = Mi.focal_length[0], Mi.focal_length[1]
        rp.screen_center[0] = 1.0 - (1.0 - Mi.principal_point[0]); rp.screen_center[1] = 1.0 - (1.0 - Mi.principal_point[1])
        for k in range(12):
            rp.camera[k] = Xi.start[k]
        rp.lens_mode = Mi.lens_mode
        for k in range(7):
            rp.lens_params[k] = Mi.lens_params[k]
        return (img, rp)
    ev, keep, kind = [], [t], "TRAINING views (the capture has no test split)"
    nv = args.eval_views
    if nv > 0 and test_transforms and os.path.exists(test_transforms):  # scripts/run.py:257-317: --test_transforms = the held-out split
        t2 = ngp.Testbed(); t2.load_training_data(test_transforms)
        d2 = t2.nerf.training.dataset
        keep.append(t2)
        for i in [int(round(k * (d2.n_images - 1) / max(nv - 1, 1))) for k in range(nv)]:
            m = d2.metadata[i]
            img = torch.from_numpy(np.ascontiguousarray(d2.image(i))).cuda()
            Mi = A.ImageMeta(); Xi = A.Xform()
            Mi.lens_mode = m.lens_mode; Mi.resolution[0], Mi.resolution[1] = m.resolution
            Mi.principal_point[0], Mi.principal_point[1] = m.principal_point; Mi.focal_length[0], Mi.focal_length[1] = m.focal_length
            for k in range(7):
                Mi.lens_params[k] = m.lens_params[k]
            for k in range(12):
                Xi.start[k] = d2.xforms[i][k]
            ev.append(view(img, Mi, Xi))
        kind = "held-out views (transforms_test.json)"
    elif nv > 0:
        for i in [int(round(k * (n - 1) / max(nv - 1, 1))) for k in range(nv)]:
            ev.append(view(images[i], M[i], X[i]))
    return dict(images=images, M=M, X=X, n=n, aabb_scale=int(d.aabb_scale), eval=ev, eval_kind=kind, name=name, keep=keep)


def load_scene(args, which=None):
    """-> dict(images=[cuda uint8 HxWx4], M, X (ctypes arrays with device pixel pointers), n, aabb_scale, eval=[(gt cuda uint8, RenderParams)], eval_kind, name, data)
    which: "synthetic" | "hard" (the two stand-ins, instant-ngp_amd/synth_scene.py) | "fox" | "lego" | a path to a transforms*.json; "auto" = lego when a copy exists, else "synthetic"."""
    which = which or args.scene
    if which == "auto":
        which = "lego" if find_lego() else "synthetic"
    if which in ("synthetic", "hard"):
        import synth_scene
        variant = "hard" if which == "hard" else "lego-format"
        images, xforms, meta, _ = synth_scene.make_dataset(args.images, args.res, "cuda", variant=variant)
        n = len(images)
        M = (A.ImageMeta * n)(); X = (A.Xform * n)()
        for i in range(n):
            M[i].pixels = images[i].data_ptr(); M[i].image_data_type = A.IMAGE_BYTE; M[i].lens_mode = A.LENS_PERSPECTIVE
            M[i].resolution[0], M[i].resolution[1] = meta["resolution"]
            M[i].principal_point[0] = M[i].principal_point[1] = 0.5
            M[i].focal_length[0], M[i].focal_length[1] = meta["focal_length"]
            for k in range(12):
                X[i].start[k] = X[i].end[k] = float(xforms[i][k])
        ev = []
        if args.eval_views > 0:
            res = args.eval_res
            gts, exf, emeta, _ = synth_scene.make_dataset(args.eval_views, res, "cuda", phase=1.234, variant=variant)
            for gt, xf in zip(gts, exf):
                rp = A.RenderParams()
                rp.resolution[0] = rp.resolution[1] = res
                rp.focal_length[0], rp.focal_length[1] = emeta["focal_length"]
                rp.screen_center[0] = rp.screen_center[1] = 0.5
                for k in range(12):
                    rp.camera[k] = float(xf[k])
                rp.lens_mode = 0
                ev.append((gt, rp))
        desc = ("hard synthetic scene (thin grille / poles / stud field / 24 small spheres, pixel-scale texture, view-dependent highlights)" if which == "hard"
                else "synthetic scene (5 boxes + 3 spheres)")
        return dict(images=images, M=M, X=X, n=n, aabb_scale=1, eval=ev, eval_kind="held-out views (synthetic test cameras)", which=which, data="synthetic",
                    metric_scene="nerf_synthetic/lego-format " + ("hard " if which == "hard" else "") + "synthetic scene",
                    name=f"NeRF nerf_synthetic/lego-format {desc} ({n} views {args.res}x{args.res} RGBA8)")
    sys.path.insert(0, os.path.join(ROOT, "instant-ngp_amd", "host"))
    import pyngp as ngp
    if which == "fox":  # the reference's real capture, through this repo's C++ loader (native JPEG decode)
        path = os.path.join(ROOT, "_ref_data", "data", "nerf", "fox", "transforms.json")
        if not os.path.exists(path):
            raise RuntimeError("--scene fox needs _ref_data/data/nerf/fox (tools/stage_reference_data.py copies it from /root/reference at build time)")
        t = ngp.Testbed(); t.load_training_data(path)
        sc = _scene_from_testbed_dataset(t, args, "")
        sc.update(which="fox", data="reference-shipped capture (data/nerf/fox)", metric_scene="data/nerf/fox",
                  name=f"NeRF data/nerf/fox real capture ({sc['n']} JPEGs {sc['M'][0].resolution[0]}x{sc['M'][0].resolution[1]}, OpenCV lens, aabb_scale {sc['aabb_scale']})")
        return sc
    path = find_lego() if which == "lego" else which
    if not path or not os.path.exists(path):
        raise RuntimeError(f"--scene {which}: no such transforms file" if which != "lego" else "--scene lego: nerf_synthetic/lego not found (NGP_LEGO_DIR, data/nerf/nerf_synthetic/lego, _ref_data/...)")
    if os.path.isdir(path):
        path = os.path.join(path, "transforms_train.json" if os.path.exists(os.path.join(path, "transforms_train.json")) else "transforms.json")
    t = ngp.Testbed(); t.load_training_data(path)
    test = os.path.join(os.path.dirname(path), "transforms_test.json") if os.path.basename(path) == "transforms_train.json" else None
    sc = _scene_from_testbed_dataset(t, args, "", test)
    is_lego = os.path.basename(os.path.dirname(os.path.abspath(path))) == "lego"
    sc.update(which="lego" if is_lego else "path", data=("nerf_synthetic/lego" if is_lego else "dataset ") + f" loaded from {path}",
              metric_scene="nerf_synthetic/lego" if is_lego else os.path.relpath(path, ROOT),
              name=f"NeRF {'nerf_synthetic/lego' if is_lego else path} ({sc['n']} views {sc['M'][0].resolution[0]}x{sc['M'][0].resolution[1]}, aabb_scale {sc['aabb_scale']})")
    return sc


def make_trainer(lib, scene, batch, rank=0, world=1, seed=1337, model_kw=None):
    cfg = A.base_model_config(scene["aabb_scale"], **(model_kw or {}))
    model = C.c_void_p()
    A.check(lib, lib.ngp_model_create(C.byref(cfg), C.c_uint64(seed), C.byref(model)))
    opts = A.default_nerf_options(scene["aabb_scale"], target_batch_size=batch, rank=rank, world_size=world, seed=seed)
    nerf = C.c_void_p()
    A.check(lib, lib.ngp_nerf_create(model, C.byref(opts), A.scene_aabb(scene["aabb_scale"]), C.byref(nerf)))
    A.check(lib, lib.ngp_nerf_set_dataset_device(nerf, scene["n"], scene["M"], scene["X"]))
    return cfg, opts, model, nerf


def get_stats(lib, nerf):
    s = A.NerfStats()
    A.check(lib, lib.ngp_nerf_get_stats(nerf, None, C.byref(s)))
    return s


def eval_psnr(lib, nerf, scene, spp=1):
    """scripts/run.py:257-317: render the evaluation views (black background, snap_to_pixel_centers, min_transmittance 1e-4, EMA
    weights, `spp` samples per pixel averaged in linear space), convert to sRGB, clip, PSNR against the view composited on black."""
    if not scene["eval"]:
        return None
    mse = []
    for gt, rp0 in scene["eval"]:
        w, h = rp0.resolution[0], rp0.resolution[1]
        frame = torch.zeros((w * h, 4), dtype=torch.float32, device="cuda")
        acc = torch.zeros_like(frame)
        rp = A.RenderParams(); C.memmove(C.byref(rp), C.byref(rp0), C.sizeof(A.RenderParams))
        rp.snap_to_pixel_centers = 1; rp.min_transmittance = 1e-4; rp.near_distance = 0.0; rp.use_inference_params = 1
        rp.render_aabb = A.scene_aabb(scene["aabb_scale"])
        for s in range(max(spp, 1)):
            rp.spp_index = s
            A.check(lib, lib.ngp_nerf_render(nerf, None, C.byref(rp), C.c_void_p(frame.data_ptr()), None))
            torch.cuda.synchronize()
            acc += (frame - acc) / float(s + 1)  # accumulate_kernel: running mean over spp (render_buffer.cu:228-260)
        lin = acc[:, :3].clamp(0, 1)
        srgb = torch.where(lin < 0.0031308, 12.92 * lin, 1.055 * lin.clamp_min(1e-12) ** (1 / 2.4) - 0.055).clamp(0, 1)
        g = gt.reshape(-1, 4).float() / 255.0
        a = g[:, 3:4]
        glin = torch.where(g[:, :3] <= 0.04045, g[:, :3] / 12.92, ((g[:, :3] + 0.055) / 1.055) ** 2.4) * a  # read_image + composite on black
        gs = torch.where(glin < 0.0031308, 12.92 * glin, 1.055 * glin.clamp_min(1e-12) ** (1 / 2.4) - 0.055).clamp(0, 1)
        mse.append(float(((srgb - gs) ** 2).mean()))
    m = sum(mse) / len(mse)
    return -10.0 * math.log10(m) if m > 0 else None


def run_ab_psnr(lib, scene, args, steps, seeds=(1337,)):
    """north_star parity proxy (the CUDA reference cannot run here): train the same scene / seed / ray stream with the production
    path and with the reference-order path, evaluate PSNR at equal step counts with the run.py procedure.  Several seeds (parameter
    initialisation AND ray stream): per path mean and standard deviation, and the PAIRED difference production - reference_order with its
    standard error -- single pairs vary by +-0.15 dB per path, so only the mean over seeds can resolve the north-star tolerance of 0.1 dB.
    --ab-clamp-variants adds both paths with mip_from_dt's crossed-bounds clamp as min(max()) (DBG_K1_MIP_CLAMP_MIN_MAX): the round-4 decision
    (tcnn's lower-bound-first form, the default) against its alternative; only multi-cascade scenes (fox) can differ."""
    ref_flags = REFERENCE_ORDER_FLAGS & ~8192 if getattr(args, "ab_lazy_k2", False) else REFERENCE_ORDER_FLAGS
    paths = [("production", 0), ("reference_order", ref_flags)]
    if args.ab_clamp_variants:
        paths += [("production_clamp_min_max", CLAMP_MIN_MAX_FLAG), ("reference_order_clamp_min_max", ref_flags | CLAMP_MIN_MAX_FLAG)]
    out = {"eval": f"{len(scene['eval'])} {scene['eval_kind']}, spp {args.eval_spp}", "steps": steps, "seeds": list(seeds), "scene": scene["name"],
           "reference_order_flags": ref_flags, "paths": {n: f for n, f in paths}}
    per = {n: {str(k): [] for k in steps} for n, _ in paths}
    wall = {n: 0.0 for n, _ in paths}
    spr = {n: [] for n, _ in paths}
    for seed in seeds:
        for name, flags in paths:
            lib.ngp_debug_set_flags(flags)
            try:
                _, _, model, nerf = make_trainer(lib, scene, args.batch, seed=seed)
                done = 0
                t0 = time.perf_counter()
                for target in steps:
                    A.check(lib, lib.ngp_nerf_train(nerf, None, target - done)); done = target
                    per[name][str(target)].append(round(eval_psnr(lib, nerf, scene, args.eval_spp), 4))
                wall[name] += time.perf_counter() - t0
                st = get_stats(lib, nerf)
                spr[name].append(round(st.measured_batch_size_before_compaction / max(st.n_rays_last, 1), 3))  # marched samples per ray at the end
                lib.ngp_nerf_destroy(nerf); lib.ngp_model_destroy(model)
            finally:
                lib.ngp_debug_set_flags(0)
    n = len(seeds)
    for name, _ in paths:
        out[name] = {k: round(float(np.mean(v)), 4) for k, v in per[name].items()}
        out[name + "_per_seed"] = per[name]
        out[name + "_std_db"] = {k: round(float(np.std(v, ddof=1)), 4) if n > 1 else None for k, v in per[name].items()}
        out[name + "_wall_s"] = round(wall[name], 2)
        out[name + "_marched_samples_per_ray_end"] = spr[name]

    def paired(x, y):
        d = {k: np.array(per[x][k]) - np.array(per[y][k]) for k in per[x]}
        r = {"delta_db": {k: round(float(v.mean()), 4) for k, v in d.items()},
             "delta_db_per_seed": {k: [round(float(e), 4) for e in v] for k, v in d.items()},
             "delta_db_stderr": {k: round(float(v.std(ddof=1) / math.sqrt(n)), 4) if n > 1 else None for k, v in d.items()}}
        r["within_0p1_db"] = {k: (abs(r["delta_db"][k]) + 2 * r["delta_db_stderr"][k] <= 0.1) if n > 1 else None for k in r["delta_db"]}
        return r
    pr = paired("production", "reference_order")
    out.update(pr)                                                                   # (keys of rounds 2-3: delta_db, delta_db_per_seed, delta_db_stderr, within_0p1_db)
    out["max_abs_delta_db"] = max(abs(v) for v in out["delta_db"].values())
    if args.ab_clamp_variants:
        out["clamp_lower_first_minus_min_max"] = {"production": paired("production", "production_clamp_min_max"), "reference_order": paired("reference_order", "reference_order_clamp_min_max")}
        out["production_minus_reference_order_with_min_max"] = paired("production_clamp_min_max", "reference_order_clamp_min_max")
    return out


def calibrate(lib=None):
    """Two seconds of box calibration, printed in `config.calibration`: boxes of this pool differ by 15-35 % on the same binary, so a line carries the
    means to tell a slow box from a regression -- device-to-device copy rate of a 1 GiB buffer (read + write bytes) and the rate of a fixed fp16 GEMM
    (8192^3 through the library GEMM: MFMA clocks)."""
    a = torch.empty(1 << 28, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
    b.copy_(a); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    copy_gbs = 8 * 2 * a.numel() * 4 / (1e6 * e0.elapsed_time(e1))
    del a, b
    x = torch.randn(8192, 8192, dtype=torch.float16, device="cuda"); y = torch.randn(8192, 8192, dtype=torch.float16, device="cuda")
    torch.matmul(x, y); torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        torch.matmul(x, y)
    e1.record(); torch.cuda.synchronize()
    gemm_tflops = 4 * 2 * 8192 ** 3 / (1e9 * e0.elapsed_time(e1))
    del x, y
    torch.cuda.empty_cache()
    extra = {}
    if lib is not None:  # shader clock seen by a light kernel: the pool's boxes come in two classes that copy / GEMM rates do not separate (W: 47 vs 70 us)
        ns = C.c_float()
        if lib.ngp_debug_clock_probe(C.byref(ns)) == 0:
            extra["valu_dependent_fma_ns"] = round(ns.value, 4)
    pr = torch.cuda.get_device_properties(0)
    extra["device"] = {"name": pr.name, "compute_units": pr.multi_processor_count, "clock_rate_khz": getattr(pr, "clock_rate", None)}
    try:  # the box's management view (power cap, clock range, performance level, partition modes): one of these may be what separates the two box classes
        import subprocess
        r = subprocess.run(["rocm-smi", "-d", "0", "--showmaxpower", "--showpower", "--showsclkrange", "--showperflevel", "--showcomputepartition", "--showmemorypartition", "--showclocks", "--json"],
                           capture_output=True, text=True, timeout=20)
        j = json.loads(r.stdout) if r.returncode == 0 and r.stdout.strip().startswith("{") else {}
        card = next(iter(j.values()), {}) if isinstance(j, dict) else {}
        extra["rocm_smi"] = {k: v for k, v in card.items() if isinstance(v, (str, int, float))} or {"raw": (r.stdout or r.stderr)[:400]}
    except Exception as e:  # (no rocm-smi, no permission, timeout: the line does not depend on it)
        extra["rocm_smi"] = {"unavailable": str(e)[:200]}
    return {"d2d_copy_GBps": round(copy_gbs, 1), "fp16_gemm_8192_TFLOPs": round(gemm_tflops, 1), **extra,
            "reference_fast_box": {"d2d_copy_GBps": 5800.0, "note": "profiles/r03_dp_diag2_comm_after_training.txt: 5.77-5.80 TB/s read+write on the box that ran 0.65 ms per step"}}


def psnr_curve_of(lib, nerf, scene, args, steps, train):
    """BASELINE.json's second metric, the run.py procedure (scripts/run.py:229-317) at the given training steps: keep training (untimed), evaluate at each step"""
    curve = {}
    for target in steps:
        cur = get_stats(lib, nerf).training_step
        if target > cur:
            train(target - cur)
        curve[str(get_stats(lib, nerf).training_step)] = round(eval_psnr(lib, nerf, scene, args.eval_spp), 3)
    return curve


def run_scene_leg(lib, args, which, pretrain, n=200):
    """Secondary leg of the default line: another scene with the same network config on 1 GPU -- `fox` (BASELINE.json configs[2], the reference's shipped real capture) or `hard` (the
    lego-hard stand-in) -- untimed load + pretrain, the same synchronize-bracketed timing as the headline over n steps, then the PSNR@step curve."""
    import copy
    a2 = copy.copy(args); a2.scene = which
    t0 = time.perf_counter()
    scene = load_scene(a2, which)
    load_s = time.perf_counter() - t0
    _, _, model, nerf = make_trainer(lib, scene, args.batch)
    curve = {}
    steps = sorted(int(x) for x in args.psnr_steps.split(",") if x) if args.psnr_steps and args.eval_views > 0 else []
    early = [k for k in steps if k <= pretrain]
    if early:  # curve points below the pretrain count are taken on the way
        curve.update(psnr_curve_of(lib, nerf, scene, args, early, lambda k: A.check(lib, lib.ngp_nerf_train(nerf, None, k))))
    cur = get_stats(lib, nerf).training_step
    A.check(lib, lib.ngp_nerf_train(nerf, None, max(pretrain - cur, 0)))  # (the occupancy grid of a real capture keeps pruning for a few thousand steps: samples marched per ray fall until then)
    torch.cuda.synchronize()
    s0 = get_stats(lib, nerf)
    t0 = time.perf_counter()
    A.check(lib, lib.ngp_nerf_train(nerf, None, n))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    s1 = get_stats(lib, nerf)
    out = {"workload": scene["name"] + ", configs/nerf/base.json, batch 2^18 samples per step", "pretrain_steps": pretrain, "steps": n, "ms_per_step": round(1e3 * el / n, 4),
           "rays_per_s": (s1.total_rays - s0.total_rays) / el, "samples_per_s": (s1.total_samples - s0.total_samples) / el,
           "rays_per_step": (s1.total_rays - s0.total_rays) / n, "samples_per_ray_compacted": (s1.total_samples - s0.total_samples) / max(s1.total_rays - s0.total_rays, 1),
           "marched_samples_per_hit_ray": s1.measured_batch_size_before_compaction / max(s1.n_rays_last, 1), "loss": s1.loss, "load_seconds": round(load_s, 2)}
    late = [k for k in steps if k > pretrain]
    if late:
        t0 = time.perf_counter()
        curve.update(psnr_curve_of(lib, nerf, scene, args, late, lambda k: A.check(lib, lib.ngp_nerf_train(nerf, None, k))))
        out["psnr_curve_seconds"] = round(time.perf_counter() - t0, 2)
    if curve:
        out["test_psnr_curve_db"] = curve
        out["test_psnr_views"] = f"{len(scene['eval'])} {scene['eval_kind']}, spp {args.eval_spp}"
    lib.ngp_nerf_destroy(nerf); lib.ngp_model_destroy(model)
    return out


def _median_ms(fn, reps=7):
    """median HIP-event time of fn() on torch's current stream == the stream the library launches on when handed stream 0 (the legacy default stream both sides use here)"""
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return float(np.median(ms))


def run_image_leg(lib):
    """BASELINE.json config 0 on the GPU (testbed_image.cu:231-302 train_image + optimizer step): data/image/albert.exr (1024^2 RGBA float, staged under _ref_data/), 2-D HashGrid
    L = 16 F = 2 T = 2^19 (every level dense), MLP 2 x 64, batch 65,536 stratified pixels per step.  Untimed: EXR decode, upload, 100 steps.  Timed: 7 x 50 steps, median."""
    exr = os.path.join(ROOT, "_ref_data", "data", "image", "albert.exr")
    if not os.path.exists(exr):
        return {"skipped": "_ref_data/data/image/albert.exr not staged (tools/stage_reference_data.py copies it from /root/reference at build time)"}
    import pyngp
    from common import ptr
    img = np.ascontiguousarray(pyngp.read_exr(exr))
    h, w = img.shape[:2]
    cfg = A.image_encmlp_config(image_resolution=max(w, h))
    hh = C.c_void_p(); A.check(lib, lib.ngp_encmlp_create(C.byref(cfg), C.c_uint64(1337), C.byref(hh)))
    n_params, n_mlp = C.c_uint64(), C.c_uint64(); lib.ngp_encmlp_n_params(hh, C.byref(n_params), C.byref(n_mlp))
    o = A.default_image_options()
    t = C.c_void_p(); A.check(lib, lib.ngp_image_create(hh, ptr(img), A.IMAGE_FLOAT, w, h, C.byref(o), C.byref(t)))
    B = int(o.batch_size)
    mse0 = C.c_float(); A.check(lib, lib.ngp_image_mse(t, 0, C.byref(mse0)))
    A.check(lib, lib.ngp_image_train(t, None, 100)); torch.cuda.synchronize()
    n = 50
    ms = _median_ms(lambda: A.check(lib, lib.ngp_image_train(t, None, n))) / n
    mse = C.c_float(); A.check(lib, lib.ngp_image_mse(t, 0, C.byref(mse)))
    # algorithmic bytes of one step (the image model's analogue of SURVEY 8d): per pixel 8 (uv) + 16 levels x 4 corners x 4 B gathered + the same entries as read-modify-write
    # scatter (x 2) + 12 (target) ; per parameter 38 (optimizer)
    step_bytes = B * (8 + 256 + 512 + 12) + BYTES_PER_PARAM_OPT * n_params.value
    out = {"workload": f"Image data/image/albert.exr ({w}x{h} RGBA f32), 2-D HashGrid L=16 F=2 T=2^19 + MLP 2x64, batch {B} pixels per step (BASELINE.json configs[0] on the GPU)",
           "steps": 7 * n, "ms_per_step": round(ms, 4), "samples_per_s": B / ms * 1e3, "n_params": int(n_params.value),
           "algorithmic_bytes_per_step": int(step_bytes), "achieved_GBps": round(step_bytes / (ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "mse_start": mse0.value, "mse_after_450_steps": mse.value, "psnr_db_after_450_steps": (round(-10 * math.log10(mse.value), 2) if mse.value > 0 else None)}
    lib.ngp_image_destroy(t); lib.ngp_encmlp_destroy(hh)
    return out


def run_sdf_leg(lib):
    """BASELINE.json config 4 (testbed_sdf.cu:1578-1631 train_sdf, :1449-1520 generate_training_samples_sdf, triangle_bvh.cu:631-660 the ray-stab ground truth): data/sdf/armadillo.obj
    (99,976 triangles, staged under _ref_data/), configs/sdf/base.json (3-D HashGrid L = 16 F = 2 T = 2^19, MLP 2 x 64, MAPE), batch 2^18 points per step.  Untimed: OBJ parse, BVH build,
    50 steps.  Timed: 16 x 20 whole steps in ONE host-clock window between two device synchronisations (sample generation + ground truth + forward / backward + optimizer; the ground truth is
    generated 16 batches per launch ahead of the steps, on a side stream: 320 timed steps consume 320 batches and generate 320), and -- on its own -- the ground truth of the half batch that goes through the BVH as one launch."""
    obj = os.path.join(ROOT, "_ref_data", "data", "sdf", "armadillo.obj")
    if not os.path.exists(obj):
        return {"skipped": "_ref_data/data/sdf/armadillo.obj not staged (tools/stage_reference_data.py copies it from /root/reference at build time)"}
    import pyngp
    from common import ptr
    tris = np.ascontiguousarray(pyngp.read_obj(obj))
    verts = tris.reshape(-1, 3).copy()
    box = A.Aabb(); scale = C.c_float()
    A.check(lib, lib.ngp_sdf_normalize_mesh_host(ptr(verts), C.c_uint64(len(verts)), C.byref(box), C.byref(scale)))
    tn = np.ascontiguousarray(verts.reshape(-1, 3, 3))
    cfg = A.sdf_encmlp_config()
    hh = C.c_void_p(); A.check(lib, lib.ngp_encmlp_create(C.byref(cfg), C.c_uint64(1337), C.byref(hh)))
    n_params, n_mlp = C.c_uint64(), C.c_uint64(); lib.ngp_encmlp_n_params(hh, C.byref(n_params), C.byref(n_mlp))
    o = A.default_sdf_options()
    t = C.c_void_p(); A.check(lib, lib.ngp_sdf_create(hh, ptr(tn), len(tn), box, C.byref(o), C.byref(t)))
    B = int(o.batch_size)
    A.check(lib, lib.ngp_sdf_train(t, None, 50)); torch.cuda.synchronize()
    # One CONTINUOUS window, host clock from a synchronised start to a synchronised end: the batches are generated ahead on a side stream, so an event pair on the caller's stream
    # around a short call would leave part of that work outside the window (it runs on while the host synchronises between repetitions).  Here the window opens with one group
    # generated before it and closes with one generated but not consumed: what lies between is the work of exactly the steps timed.
    n, reps = 20, 16
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        A.check(lib, lib.ngp_sdf_train(t, None, n))
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / (n * reps)
    loss = C.c_float(); A.check(lib, lib.ngp_sdf_loss(t, None, C.byref(loss)))
    # the ground truth alone: the near-surface 3/8 + uniform 1/8 of the last batch are the points generate_training_samples_sdf hands to the BVH (the 4/8 on the surface have distance 0)
    pp, dp = C.c_void_p(), C.c_void_p(); lib.ngp_sdf_batch_ptrs(t, C.byref(pp), C.byref(dp))
    n_exact = B // 8 * 4
    n_bvh = B - n_exact
    dist_out = torch.zeros(n_bvh, dtype=torch.float32, device="cuda")
    pos_ptr = C.c_void_p(pp.value + n_exact * 12)
    ms_gt = _median_ms(lambda: A.check(lib, lib.ngp_sdf_signed_distance(t, None, pos_ptr, n_bvh, C.c_void_p(dist_out.data_ptr()))))
    iou = C.c_double(); A.check(lib, lib.ngp_sdf_iou(t, 1 << 18, C.byref(iou)))
    step_bytes = B * (12 + 512 + 1024 + 4) + BYTES_PER_PARAM_OPT * n_params.value   # per point 12 (position) + 16 levels x 8 corners x 4 B gathered + RMW scatter + 4 (distance); 38 per parameter
    out = {"workload": f"SDF data/sdf/armadillo.obj ({len(tn)} triangles), configs/sdf/base.json (3-D HashGrid L=16 F=2 T=2^19 + MLP 2x64, MAPE), batch {B} points per step (BASELINE.json configs[4])",
           "steps": reps * n, "ms_per_step": round(ms, 4), "samples_per_s": B / ms * 1e3,
           "ms_ground_truth_alone": round(ms_gt, 4), "ground_truth_points": n_bvh, "inside_fraction": round(float((dist_out < 0).float().mean().item()), 4),
           "batches_ahead": int(os.environ.get("NGP_SDF_GROUP", "16")) if not int(os.environ.get("NGP_SDF_NO_PREFETCH", "0") or 0) else 0,
           "split_note": "ground truth = unsigned BVH distance + up to 32 stab rays for the near-surface and uniform half of a batch, timed on its own on the last batch's points as ONE launch (without the trainer's upper bounds, which only prune more).  The trainer generates samples + ground truth of `batches_ahead` batches per launch on a side stream, ahead of the steps that train on them (a launch lasts as long as its longest walk: 1 / 2 / 4 batches 2.2 / 2.6 / 4.0 ms, profiles/r06_exp_sdf_multibatch.jsonl); every timed step still consumes one freshly generated batch, so ms_per_step holds sample generation + ground truth + forward / backward + scatter + optimizer",
           "n_params": int(n_params.value), "algorithmic_bytes_per_step": int(step_bytes), "achieved_GBps": round(step_bytes / (ms * 1e-3) / 1e9, 1),
           "frac_of_hbm_peak": round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "loss_mape": loss.value, "iou_after_370_steps": round(iou.value, 4)}
    lib.ngp_sdf_destroy(t); lib.ngp_encmlp_destroy(hh)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pretrain", type=int, default=1010, help="untimed training steps that bring the occupancy grid to steady state (1010 + the driver's 5 warm-up steps: the 20 timed steps then hold ONE occupancy-grid update -- the reference updates every 16th step, i.e. 1.25 per 20 steps; the line says how many its window held and carries a 200-step figure beside it)")
    ap.add_argument("--scene", default="auto", help="auto (nerf_synthetic/lego when a copy exists -- NGP_LEGO_DIR, data/nerf/nerf_synthetic/lego, _ref_data/... --, else the synthetic stand-in) | synthetic | hard | fox | lego | path to a transforms*.json (scripts/scenes.py:25-32)")
    ap.add_argument("--steady-steps", type=int, default=200, help="a second, longer timed window behind the K timed steps (reported beside them; 0 = off)")
    ap.add_argument("--no-hard-leg", action="store_true", help="skip the lego-hard stand-in leg of the default line")
    ap.add_argument("--images", type=int, default=100)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--batch", type=int, default=1 << 18)
    ap.add_argument("--profile-steps", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fox-leg", action="store_true", help="skip the secondary data/nerf/fox leg of the default line")
    ap.add_argument("--no-f4-legs", action="store_true", help="skip the image (BASELINE.json configs[0]) and SDF (configs[4]) legs of the default line")
    ap.add_argument("--no-calibration", action="store_true")
    ap.add_argument("--eval-views", type=int, default=4, help="views rendered (untimed) for PSNR, run.py --test_transforms procedure")
    ap.add_argument("--eval-res", type=int, default=400)
    ap.add_argument("--eval-spp", type=int, default=1)
    ap.add_argument("--psnr-steps", type=str, default="1000,5000,10000,35000", help="comma separated training steps at which the PSNR is evaluated after the timed region (untimed; BASELINE.json's PSNR@step, run.py's default n_steps = 35000); '' = off")
    ap.add_argument("--ab-psnr", type=str, default="", help="comma separated steps: equal-step PSNR of the production path vs the reference-order path (two fresh trainings, untimed)")
    ap.add_argument("--ab-seeds", type=int, default=1, help="--ab-psnr: number of seeds (1337, 1338, ...) per path; mean, standard deviation and the paired difference with its standard error are reported")
    ap.add_argument("--ab-seed0", type=int, default=1337, help="--ab-psnr: first seed")
    ap.add_argument("--ab-only", action="store_true", help="run only the --ab-psnr experiment (no timed region) and print its JSON: tools/ab_psnr_parallel.py runs several of these side by side, one seed range each")
    ap.add_argument("--ab-lazy-k2", action="store_true", help="--ab-psnr: the reference-order path keeps the lazy K2 (it produces the same compacted batch as the eager order: tests/test_gpu_train.py::test_lazy_k2_matches_eager) -- the path then differs from production in sample positions (sequential K1), compositing order (sequential K3) and gradient sums (half atomics) only, at 2/3 of the time")
    ap.add_argument("--ab-clamp-variants", action="store_true", help="--ab-psnr: also train both paths with mip_from_dt's crossed-bounds clamp as min(max()) (ablation of the round-4 decision)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="N > 1: weak = --batch samples per GPU per step (default, the driver's mode); strong = --batch samples per step in total (B / N per GPU)")
    ap.add_argument("--dp-backend", choices=["auto", "rccl", "torch"], default="auto", help="N > 1: gradient / counter all-reduce inside libngp_hip (RCCL, ngp_comm_*) or through torch.distributed")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON result): libraries that print banners to fd 1 (RCCL's version block at init) go to stderr
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    force_dp = os.environ.get("NGP_FORCE_DP", "0") == "1"  # exercise the multi-GPU step (all-reduce of size 1) on one GPU
    if world > 1 or force_dp or os.environ.get("NGP_BENCH_INIT_PG", "0") == "1":  # NGP_BENCH_INIT_PG: diagnostic -- a process group without a data-parallel step
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run"

    lib = A.load_hip()  # raises if libngp_hip.so is missing: no CPU fallback on the product path
    A.check(lib, lib.ngp_init())  # the library's helper streams exist before the first collective creates an RCCL communicator
    if os.environ.get("NGP_DEBUG_FLAGS"):  # ablation switches of ngp_kernels.hpp (default 0 = production path)
        lib.ngp_debug_set_flags(int(os.environ["NGP_DEBUG_FLAGS"], 0))
    assert lib.ngp_device_available() == 1

    calibration = None if args.no_calibration else calibrate(lib)
    # what is NOT the production path must be visible in the line: ablation bits in effect (ngp_debug_set_flags / NGP_DEBUG_FLAGS_OR) and every NGP_* environment knob
    lib.ngp_debug_get_flags.restype = C.c_uint32
    ablation = {"debug_flags": int(lib.ngp_debug_get_flags()), "env_knobs": {k: v for k, v in sorted(os.environ.items()) if k.startswith("NGP_")}}
    scene = load_scene(args)
    if args.ab_only:
        ab = run_ab_psnr(lib, scene, args, sorted(int(x) for x in args.ab_psnr.split(",") if x), seeds=[args.ab_seed0 + i for i in range(max(args.ab_seeds, 1))])
        result_out.write(json.dumps(ab) + "\n"); result_out.flush()
        return
    if args.scaling == "strong":  # total work fixed: every rank trains B / N samples per step (the library needs a multiple of 256)
        args.batch = max(256, args.batch // world // 256 * 256)
    cfg, opts, model, nerf = make_trainer(lib, scene, args.batch, rank, world)
    n_params, n_mlp = C.c_uint64(), C.c_uint64()
    lib.ngp_model_n_params(model, C.byref(n_params), C.byref(n_mlp))

    # ---- data-parallel step: inside the library over RCCL (ngp_comm_*), or through torch.distributed on zero-copy views -------------
    dp = world > 1 or force_dp
    dp_backend = None
    grad_view = cnt_view = None
    if dp:
        if args.dp_backend in ("auto", "rccl") and hasattr(lib, "ngp_comm_init"):
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                buf = (C.c_uint8 * 128)()
                A.check(lib, lib.ngp_comm_unique_id(buf))
                uid = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()
            uid = uid.cuda(); dist.broadcast(uid, 0); uid = uid.cpu()
            rc = lib.ngp_comm_init(nerf, rank, world, (C.c_uint8 * 128)(*uid.tolist()))
            ok = torch.tensor([1 if rc == 0 else 0], device="cuda"); dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                dp_backend = "rccl-in-library"
            else:
                if rc == 0:
                    lib.ngp_comm_destroy(nerf)  # another rank failed: no rank may keep a communicator while the torch.distributed path runs the step
                if args.dp_backend == "rccl":
                    raise RuntimeError("ngp_comm_init failed on some rank: " + lib.ngp_last_error().decode())
        if dp_backend is None:
            dp_backend = "torch.distributed"
            g = C.c_void_p(); lib.ngp_model_param_ptrs(model, None, None, None, C.byref(g))
            grad_view = torch.as_tensor(CudaView(g.value, n_params.value, "<f2"), device="cuda")
            cp = C.c_void_p(); lib.ngp_nerf_counter_ptrs(nerf, C.byref(cp))
            cnt_view = torch.as_tensor(CudaView(cp.value, 3, "<i4"), device="cuda")  # {marched, compacted, loss sum in units of 2^-24}

    host_s = {}  # torch.distributed path: host-side enqueue time per call (to tell a host-bound loop from a device-bound one)

    def step(n=1):
        if not dp or dp_backend == "rccl-in-library":
            A.check(lib, lib.ngp_nerf_train(nerf, None, n))  # multi-rank: the library all-reduces counters and gradient buckets itself
            return
        pc = time.perf_counter
        for _ in range(n):
            t0 = pc()
            A.check(lib, lib.ngp_nerf_train_prep(nerf, None))
            A.check(lib, lib.ngp_nerf_train_forward(nerf, None))    # K1 (unless pre-launched) .. K4
            t1 = pc()
            dist.all_reduce(cnt_view)    # three uint32 so that every rank derives the same next rays_per_batch and reports the union batch's loss
            t2 = pc()
            A.check(lib, lib.ngp_nerf_train_backward(nerf, None))   # controller, next step's K1 on its own stream, T1 / scatter / W
            t3 = pc()
            dist.all_reduce(grad_view)   # RCCL sum of hash-grid + MLP gradients (fp16, 23.4 MB) over xGMI, overlapped by the next K1
            t4 = pc()
            A.check(lib, lib.ngp_nerf_train_finish(nerf, None))     # optimizer
            t5 = pc()
            for k, v in (("prep+forward", t1 - t0), ("allreduce_counters", t2 - t1), ("backward", t3 - t2), ("allreduce_gradients", t4 - t3), ("finish", t5 - t4)):
                host_s[k] = host_s.get(k, 0.0) + v
            host_s["steps"] = host_s.get("steps", 0) + 1

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    step(args.pretrain)
    step(args.warmup)
    barrier()
    host_s.clear()
    s0 = get_stats(lib, nerf)
    t0 = time.perf_counter()
    step(args.steps)
    host_ms = {k: round(1e3 * v / max(host_s.get("steps", 1), 1), 4) for k, v in host_s.items() if k != "steps"}  # snapshot of the timed region only
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
        dist.barrier()
    s1 = get_stats(lib, nerf)
    rays = s1.total_rays - s0.total_rays                # global rays launched over the K steps (all ranks)
    samples = (s1.total_samples - s0.total_samples) * world
    value = rays / elapsed
    # occupancy-grid updates inside the K timed steps (training_prep_nerf at testbed.cu:4596's cadence: every 16th step beyond step 256): the step counter before / after says how many
    def n_grid_updates(a, b):  # prep of step k updates when k % 16 == 0 (k >= 256)
        return sum(1 for k in range(a, b) if k >= 256 and k % 16 == 0)
    window = {"first_step": int(s0.training_step), "steps": args.steps, "grid_updates_in_window": n_grid_updates(int(s0.training_step), int(s1.training_step)),
              "grid_updates_per_window_steady_state": round(args.steps / 16.0, 3)}
    steady = None
    if args.steady_steps > 0:  # a longer window behind the K steps: the steady-state average (same bracketing)
        barrier()
        t0 = time.perf_counter()
        step(args.steady_steps)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        if dist is not None:
            e2 = torch.tensor([el2], dtype=torch.float64, device="cuda"); dist.all_reduce(e2, op=dist.ReduceOp.MAX); el2 = float(e2.item()); dist.barrier()
        s2 = get_stats(lib, nerf)
        steady = {"steps": args.steady_steps, "ms_per_step": round(1e3 * el2 / args.steady_steps, 4), "rays_per_s": (s2.total_rays - s1.total_rays) / el2,
                  "samples_per_s": (s2.total_samples - s1.total_samples) * world / el2, "grid_updates_in_window": n_grid_updates(int(s1.training_step), int(s2.training_step))}

    # ---- roofline leg: per-kernel HIP-event timing over further (untimed) steps --------------------------------
    lib.ngp_profile_enable(1)
    step(max(args.profile_steps, 1))
    args.profile_steps = max(args.profile_steps, 1)
    npf = lib.ngp_profile_count()
    ms = (C.c_double * npf)(); cnt = (C.c_uint64 * npf)()
    lib.ngp_profile_read(ms, cnt)
    lib.ngp_profile_enable(0)
    s3 = get_stats(lib, nerf)
    lib.ngp_profile_name.restype = C.c_char_p
    kern = {lib.ngp_profile_name(i).decode(): (ms[i], cnt[i]) for i in range(npf) if cnt[i]}
    n_inf_avg = s3.network_evaluations  # network evaluations of the last step's K2 (lazy K2: fewer than the marched samples)
    stash = bool(lib.ngp_nerf_uses_k2_stash(nerf)) if hasattr(lib, "ngp_nerf_uses_k2_stash") else True
    t1_bytes = BYTES_PER_SAMPLE_T1 if stash else BYTES_PER_SAMPLE_T1_GATHER
    # algorithmic bytes per launch of the kernels that have a byte model.  Each byte is charged ONCE: the 512-byte gather belongs to K2,
    # the scatter unit (T1 + k_grad_bin + k_grad_accumulate) is charged what it moves itself.
    # Single-GPU production step: k_grad_accumulate applies the optimizer to the hashed levels in its epilogue, k_optimizer sweeps the MLP + dense levels only -- each kernel is
    # charged the parameters it updates (38 B each), the step's sum is unchanged.
    lib.ngp_model_last_sweep_params.restype = C.c_uint64
    n_sweep = int(lib.ngp_model_last_sweep_params(model)) or n_params.value
    n_fused = n_params.value - n_sweep
    per_launch_bytes = {"k_inference": BYTES_PER_SAMPLE_FWD * n_inf_avg, "k_train_fwd_bwd+k_grad_bin+k_grad_accumulate": t1_bytes * args.batch + BYTES_PER_PARAM_OPT * n_fused,
                        "k_optimizer": BYTES_PER_PARAM_OPT * n_sweep}
    fused_ms = None
    if "k_train_fused" in kern:
        # round 5: T1 and W are one kernel (k_train_fused); it is the front of the scatter unit (it emits dL/d(enc)) AND the weight-gradient kernel
        fused_ms = kern["k_train_fused"][0] / kern["k_train_fused"][1]
        kern["k_train_fwd_bwd"] = kern.pop("k_train_fused")
    if "k_grad_bin+accumulate" in kern and "k_train_fwd_bwd" in kern:
        # the scatter runs in two follow-up kernels of T1: one unit of algorithmic work, one entry of the per-step table
        t1 = kern.pop("k_train_fwd_bwd"); gb = kern.pop("k_grad_bin+accumulate")
        kern["k_train_fwd_bwd+k_grad_bin+k_grad_accumulate"] = (t1[0] + gb[0], t1[1])
    elif "k_train_fwd_bwd" in kern:
        kern["k_train_fwd_bwd+k_grad_bin+k_grad_accumulate"] = kern.pop("k_train_fwd_bwd")
    kern_ms = {k: v[0] / args.profile_steps for k, v in kern.items()}
    # the dominant SINGLE kernel with a byte model: K2 (k_inference_tiles) or the optimizer sweep -- the scatter unit is three kernels and is reported beside it
    single = [k for k in ("k_inference", "k_optimizer") if k in kern]
    dominant = max(single, key=lambda k: kern[k][0] / kern[k][1])
    avg_ms = kern[dominant][0] / kern[dominant][1]
    achieved = per_launch_bytes[dominant] / (avg_ms * 1e-3) / 1e9
    pmc = None
    for fn in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):  # PMC counters cannot be sampled from inside the process: committed rocprofv3 --pmc passes (tools/pmc_traffic.sh)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", fn)))
            traffic_src = f"profiles/{fn} (separate rocprofv3 --pmc passes of this command; see the file for the correction applied)"
            break
        except Exception:
            pass

    def pmc_bytes(k):
        try:
            return pmc[k]["bytes_per_launch"]
        except Exception:
            return None
    traffic = pmc_bytes(dominant)
    if traffic is None:
        traffic_src = None
    ms_step = 1e3 * elapsed / args.steps
    su = "k_train_fwd_bwd+k_grad_bin+k_grad_accumulate"
    units = {}
    for k in (dominant, su, "k_optimizer", "k_inference"):
        if k in kern and k not in units:
            t_ms = kern[k][0] / kern[k][1]
            units[k] = {"algorithmic_bytes_per_launch": int(per_launch_bytes[k]), "avg_launch_ms": round(t_ms, 4), "achieved_GBps": round(per_launch_bytes[k] / (t_ms * 1e-3) / 1e9, 1),
                        "frac": round(per_launch_bytes[k] / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": pmc_bytes(k),
                        "traffic_over_algorithmic": (round(pmc_bytes(k) / per_launch_bytes[k], 3) if pmc_bytes(k) else None)}
    # whole step: every algorithmic byte of the step (each charged once) over the measured step time of the timed region
    step_bytes = BYTES_PER_SAMPLE_FWD * n_inf_avg + t1_bytes * args.batch + BYTES_PER_PARAM_OPT * n_params.value + BYTES_PER_SAMPLE_K3 * n_inf_avg
    whole = {"algorithmic_bytes_per_step": int(step_bytes), "ms_per_step": round(ms_step, 4), "achieved_GBps": round(step_bytes / (ms_step * 1e-3) / 1e9, 1),
             "frac": round(step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             "note": "K2 548 B x evaluations + scatter unit (28 + 64 + 8 + 1024) B x batch + optimizer 38 B x parameters (hashed levels: inside k_grad_accumulate's epilogue on one GPU, charged to the scatter unit; MLP + dense levels: k_optimizer) + K3/K4 38 B x evaluations; K1 (VALU bound lattice march), the occupancy-grid update and W (MFMA) move no modelled bytes"}
    flop_step = FLOP_PER_SAMPLE_FWD * n_inf_avg + FLOP_PER_SAMPLE_TRAIN * args.batch
    wg = kern.get("k_wgrad")
    fused_mfma = ({"kernel": "k_train_fused (T1 + W: forward, dgrad, weight gradients of one batch)", "algorithmic_flop": int(FLOP_PER_SAMPLE_TRAIN * args.batch), "avg_launch_ms": round(fused_ms, 4),
                   "achieved_tflops": round(FLOP_PER_SAMPLE_TRAIN * args.batch / (fused_ms * 1e-3) / 1e12, 2),
                   "frac_of_2.5PF": round(FLOP_PER_SAMPLE_TRAIN * args.batch / (fused_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)} if fused_ms else None)
    mfma = {"flop_per_step": int(flop_step), "achieved_tflops": round(flop_step / (ms_step * 1e-3) / 1e12, 2), "peak_tflops": MFMA_PEAK_TFLOPS,
            "frac_of_2.5PF": round(flop_step / (ms_step * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "k_wgrad": ({"algorithmic_flop": int(FLOP_PER_SAMPLE_FWD * args.batch), "avg_launch_ms": round(wg[0] / wg[1], 4),
                         "achieved_tflops": round(FLOP_PER_SAMPLE_FWD * args.batch / (wg[0] / wg[1] * 1e-3) / 1e12, 2),
                         "frac_of_2.5PF": round(FLOP_PER_SAMPLE_FWD * args.batch / (wg[0] / wg[1] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)} if wg else None),
            **({"k_train_fused": fused_mfma} if fused_mfma else {}),
            "note": "the path is gather / scatter bound: MFMA is a minor term (SURVEY 8d); counter-based MFMA busy fractions: profiles/r02_pmc_mfma*.txt"}
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(per_launch_bytes[dominant]),
                "units": units, "whole_step": whole,
                "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(kern_ms.items(), key=lambda kv: -kv[1])}, "mfma": mfma}

    t_eval = time.perf_counter()
    psnr = eval_psnr(lib, nerf, scene, args.eval_spp) if (rank == 0 and args.eval_views > 0) else None
    t_eval = time.perf_counter() - t_eval
    psnr_step = get_stats(lib, nerf).training_step

    # optional PSNR@step curve (BASELINE.json's second metric): keep training, untimed, and evaluate at the requested steps
    psnr_curve = {}
    t_curve = time.perf_counter()
    if rank == 0 and world == 1 and args.psnr_steps and args.eval_views > 0:
        if psnr is not None:
            psnr_curve[str(psnr_step)] = round(psnr, 3)
        psnr_curve.update(psnr_curve_of(lib, nerf, scene, args, [k for k in sorted(int(x) for x in args.psnr_steps.split(",") if x) if k > psnr_step], step))
    t_curve = time.perf_counter() - t_curve

    # ---- CPU baseline: the oracle (port) runs bounded steps from the same trained state --------------------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(lib, model, nerf, cfg, scene, s3, args)

    fox_leg = hard_leg = None
    main_default = scene.get("which") in ("synthetic", "lego")
    if rank == 0 and world == 1 and main_default and not args.no_fox_leg:
        try:
            fox_leg = run_scene_leg(lib, args, "fox", 5020)
        except Exception as e:  # the capture is staged from /root/reference at build time; a box without it still reports the headline
            fox_leg = {"skipped": str(e)[:200]}
    if rank == 0 and world == 1 and main_default and not args.no_hard_leg:
        try:
            hard_leg = run_scene_leg(lib, args, "hard", 2020)
        except Exception as e:
            hard_leg = {"skipped": str(e)[:200]}

    f4_legs = {}
    if rank == 0 and world == 1 and main_default and not args.no_f4_legs:
        sys.path.insert(0, os.path.join(ROOT, "instant-ngp_amd", "host"))
        for name, fn in (("image", run_image_leg), ("sdf", run_sdf_leg)):
            t_leg = time.perf_counter()
            try:
                f4_legs[name] = fn(lib)
            except Exception as e:  # a leg that cannot run (data not staged, pyngp not built) never costs the headline
                f4_legs[name] = {"skipped": str(e)[:200]}
            f4_legs[name]["leg_wall_seconds"] = round(time.perf_counter() - t_leg, 2)

    ab = None
    if rank == 0 and world == 1 and args.ab_psnr:
        ab = run_ab_psnr(lib, scene, args, sorted(int(x) for x in args.ab_psnr.split(",") if x), seeds=[args.ab_seed0 + i for i in range(max(args.ab_seeds, 1))])

    if rank == 0:
        out = {
            # rays/s = samples/s / (samples per ray), and the second factor is a property of the SCENE (4.5 on the easy stand-in, ~60 on fox): the portable figure is samples/s
            "metric": f"training rays/sec on {scene['metric_scene']}, configs/nerf/base.json, B=2^18 samples/step ({samples / elapsed / 1e6:.0f} M samples/s at {samples / max(rays, 1):.2f} compacted samples per ray)",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f16", "data": scene["data"],
            "config": {"workload": scene["name"] + ", configs/nerf/base.json (hash L=8 F=4 T=2^19, MLP 64), batch " + (f"2^{int(math.log2(args.batch))}" if args.batch & (args.batch - 1) == 0 else str(args.batch)) + " samples per GPU per step, rays/step adaptive (cap 2^18)",
                       "parallelism": f"dp{world}", **({"dp_backend": dp_backend} if dp_backend else {}),
                       "pretrain_steps": args.pretrain, "timed_window": window, **({"steady_window": steady} if steady else {}),
                       "rays_per_step": rays / args.steps, "samples_per_s": samples / elapsed,
                       "samples_per_ray_compacted": samples / max(rays, 1),
                       "ray_hit_fraction": s1.n_rays_last / max(s1.rays_per_batch, 1), "rays_hit_last_step": s1.n_rays_last, "rays_per_batch_last_step": s1.rays_per_batch,
                       "marched_samples_last_step": s1.measured_batch_size_before_compaction, "network_evaluations_last_step": s1.network_evaluations,
                       "loss": s1.loss, "train_psnr_estimate_db": (-10 * math.log10(s1.loss) if s1.loss > 0 else None), "training_step_end": s1.training_step,
                       "test_psnr_db": psnr, "test_psnr_at_step": (psnr_step if psnr is not None else None),
                       "test_psnr_views": f"{len(scene['eval'])} {scene['eval_kind']}, spp {args.eval_spp}", "test_psnr_eval_seconds": round(t_eval, 3),
                       **({"calibration": calibration} if calibration else {}),
                       "production_path": ablation["debug_flags"] == 0 and not ablation["env_knobs"], **({"ablation": ablation} if (ablation["debug_flags"] or ablation["env_knobs"]) else {}),
                       **({"dp_host_enqueue_ms_per_step": host_ms} if host_ms else {}),
                       **({"test_psnr_curve_db": psnr_curve, "test_psnr_curve_seconds": round(t_curve, 2)} if psnr_curve else {}),
                       **({"ab_psnr": ab} if ab else {})},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
            **({"legs": {**({"fox": fox_leg} if fox_leg else {}), **({"hard": hard_leg} if hard_leg else {}), **f4_legs}} if (fox_leg or hard_leg or f4_legs) else {}),
        }
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if dist is not None:
        dist.destroy_process_group()


def run_cpu_baseline(lib, model, nerf, cfg, scene, st, args):
    """Time the CPU oracle (a port: the reference has no CPU path) on a bounded sample of the same workload:
    training steps at B_cpu = 2^15 samples starting from the GPU's trained parameters and occupancy grid."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    from common import OraModel, ptr
    ora = oracle_py.load()
    B_cpu = 1 << 15
    n_img, M, X = scene["n"], scene["M"], scene["X"]
    om = OraModel(ora, cfg)
    p = np.empty(om.n, dtype=np.float32)
    A.check(lib, lib.ngp_model_get_params_host(model, ptr(p), C.c_uint64(p.size)))
    om.params_fp[:] = p
    ora.ora_model_sync_half(om.h)
    o2 = A.default_nerf_options(scene["aabb_scale"], target_batch_size=B_cpu)
    ot = C.c_void_p()
    assert ora.ora_nerf_create(om.h, C.byref(o2), A.scene_aabb(scene["aabb_scale"]), C.byref(ot)) == 0
    host_imgs = [im.cpu().numpy() for im in scene["images"]]
    Mh = (A.ImageMeta * n_img)()
    for i in range(n_img):
        C.memmove(C.byref(Mh[i]), C.byref(M[i]), C.sizeof(A.ImageMeta))
        Mh[i].pixels = host_imgs[i].ctypes.data
    ora.ora_nerf_set_dataset(ot, n_img, Mh, X)
    n_casc = int(o2.max_cascade) + 1
    gp = C.c_void_p(); lib.ngp_nerf_density_grid_ptrs(nerf, C.byref(gp), None, None)
    grid = torch.as_tensor(CudaView(gp.value, 128 ** 3 * n_casc, "<f4"), device="cuda").cpu().numpy()
    C.memmove(ora.ora_nerf_density_grid(ot), grid.ctypes.data, grid.nbytes)
    ora.ora_nerf_update_mean_and_bitfield(ot)
    rays_cpu = max(256, int(st.rays_per_batch * B_cpu / args.batch) // 256 * 256)
    ora.ora_nerf_set_rays_per_batch(ot, rays_cpu)
    # repeat the step until ~10 s of wall time are spent (bounded: at most 32 steps); the controller adapts rays/step like on the GPU
    n_steps, rays_total, t0 = 0, 0, time.perf_counter()
    while n_steps < 32 and (n_steps == 0 or time.perf_counter() - t0 < 10.0):
        s = A.NerfStats(); ora.ora_nerf_get_stats(ot, C.byref(s))
        rays_total += s.rays_per_batch if n_steps else rays_cpu
        assert ora.ora_nerf_train_forward_backward(ot) == 0
        assert ora.ora_nerf_train_finish(ot) == 0
        n_steps += 1
    dt = time.perf_counter() - t0
    s = A.NerfStats(); ora.ora_nerf_get_stats(ot, C.byref(s))
    ora.ora_nerf_destroy(ot)
    # forward-only legs of SURVEY.md 8(d): NeRF network on 2^18 ray-coherent samples, image model (config 0) on its 65,536-sample batch
    legs = {}
    from common import random_coords
    c = random_coords(1 << 18, seed=1, ray_coherent=True)
    t1 = time.perf_counter(); om.inference(c); d1 = time.perf_counter() - t1
    legs["nerf_forward"] = {"value": (1 << 18) / d1, "unit": "samples/s", "sample": f"2^18 samples, {d1:.2f} s"}
    ic = A.image_encmlp_config(); ih = C.c_void_p()
    if ora.ora_encmlp_create(C.byref(ic), C.c_uint64(1337), C.byref(ih)) == 0:
        uv = np.random.default_rng(1).random((65536, 2), dtype=np.float32); o16 = np.zeros((65536, 3), np.uint16)
        t2 = time.perf_counter(); ora.ora_encmlp_inference(ih, ptr(uv), 2, 65536, ptr(o16), 3); d2 = time.perf_counter() - t2
        legs["image_forward"] = {"value": 65536 / d2, "unit": "samples/s", "sample": f"config 0 model (2-D HashGrid L16 F2 T=2^19 + MLP 2x64), 65,536 samples, {d2:.3f} s"}
        ora.ora_encmlp_destroy(ih)
    return {"value": rays_total / dt, "unit": "rays/s", "cores": int(ora.ora_num_threads()), "kind": "port",
            "sample": f"{n_steps} training steps (K1..K6, occupancy prep excluded) at B=2^15 samples, {rays_total} rays, last step {s.measured_batch_size} compacted samples, "
                      f"from the GPU's trained state; OpenMP over {ora.ora_num_threads()} threads; {dt:.2f} s",
            "legs": legs}


if __name__ == "__main__":
    main()
