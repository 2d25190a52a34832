#!/usr/bin/env python
"""Rates of the image and SDF primitives (BASELINE.json configs 0 and 4) on the GPU box.

SDF (config 4): armadillo.obj when the reference's data is staged under _ref_data/ (an icosphere otherwise), configs/sdf/base.json model, batch 2^18:
  * ground truth alone: ngp_sdf_signed_distance over the near-surface + uniform half of a training batch (what generate_training_samples_sdf hands to the BVH)
  * whole training step (sample generation + ground truth + fused forward / backward + optimizer)
image (config 0): a synthetic 2048 x 2048 float image, configs/image/base.json model with T = 2^19, batches of 2^16 and 2^18 pixels.
One JSON line per case (median of HIP-event timings).  Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel table.
usage: python tools/f4_bench.py"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "instant-ngp_amd"), os.path.join(ROOT, "instant-ngp_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ngp_abi as A  # noqa: E402
from common import ptr  # noqa: E402


def timed(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return float(np.median(ms))


def icosphere(subdiv=5, radius=0.4):
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(x, np.float64) / np.linalg.norm(x) for x in v]
    for _ in range(subdiv):
        nf = []; cache = {}
        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]; v.append(m / np.linalg.norm(m)); cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    v = np.array(v) * radius + 0.5
    return np.ascontiguousarray(v[np.array(f)].astype(np.float32))


def sdf_cases(lib):
    obj = os.path.join(ROOT, "_ref_data", "data", "sdf", "armadillo.obj")
    if os.path.exists(obj):
        import pyngp
        tris, name = np.ascontiguousarray(pyngp.read_obj(obj)), "data/sdf/armadillo.obj"
    else:
        tris, name = icosphere(), "icosphere (the reference's mesh is not staged)"
    verts = tris.reshape(-1, 3).copy()
    box = A.Aabb(); scale = C.c_float()
    A.check(lib, lib.ngp_sdf_normalize_mesh_host(ptr(verts), C.c_uint64(len(verts)), C.byref(box), C.byref(scale)))
    tn = np.ascontiguousarray(verts.reshape(-1, 3, 3))
    cfg = A.sdf_encmlp_config()
    hh = C.c_void_p(); A.check(lib, lib.ngp_encmlp_create(C.byref(cfg), C.c_uint64(1337), C.byref(hh)))
    o = A.default_sdf_options()
    t = C.c_void_p(); A.check(lib, lib.ngp_sdf_create(hh, ptr(tn), len(tn), box, C.byref(o), C.byref(t)))
    B = o.batch_size
    A.check(lib, lib.ngp_sdf_train(t, None, 50)); torch.cuda.synchronize()
    # one continuous host-clock window between two synchronisations (the batches are generated ahead on a side stream: an event pair around a short call would leave part of that
    # work outside the window -- it runs on while the host synchronises between repetitions)
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(16):
        A.check(lib, lib.ngp_sdf_train(t, None, 20))
    torch.cuda.synchronize(); ms_step = (time.perf_counter() - t0) * 1e3 / 320
    print(json.dumps({"op": "sdf training step (samples + BVH ground truth + fwd / bwd + optimizer)", "mesh": name, "triangles": len(tn), "batch": B, "ms": round(ms_step, 4), "samples_per_s": B / ms_step * 1e3}), flush=True)
    # the ground truth alone on the points a batch hands to the BVH (near-surface 3/8 + uniform 1/8 of the batch), without the upper bounds (an upper bound only prunes more)
    pp, dp = C.c_void_p(), C.c_void_p(); lib.ngp_sdf_batch_ptrs(t, C.byref(pp), C.byref(dp))
    n_exact = B // 8 * 4
    n = B - n_exact
    out = torch.zeros(n, dtype=torch.float32, device="cuda")
    pos_ptr = C.c_void_p(pp.value + n_exact * 12)
    ms = timed(lambda: A.check(lib, lib.ngp_sdf_signed_distance(t, None, pos_ptr, n, C.c_void_p(out.data_ptr()))))
    inside = float((out < 0).float().mean().item())
    print(json.dumps({"op": "sdf ground truth (unsigned distance + 32 stab rays) of one batch's near-surface and uniform points", "mesh": name, "points": n, "inside_fraction": round(inside, 4), "ms": round(ms, 4),
                      "points_per_s": n / ms * 1e3, "ms_per_2^18_points": round(ms * (1 << 18) / n, 4)}), flush=True)
    u = torch.rand((1 << 18, 3), device="cuda") * 0.8 + 0.1
    out2 = torch.zeros(1 << 18, dtype=torch.float32, device="cuda")
    ms = timed(lambda: A.check(lib, lib.ngp_sdf_signed_distance(t, None, C.c_void_p(u.data_ptr()), 1 << 18, C.c_void_p(out2.data_ptr()))))
    print(json.dumps({"op": "sdf ground truth of 2^18 uniform points in [0.1, 0.9]^3", "mesh": name, "inside_fraction": round(float((out2 < 0).float().mean().item()), 4), "ms": round(ms, 4), "points_per_s": (1 << 18) / ms * 1e3}), flush=True)
    lib.ngp_sdf_destroy(t); lib.ngp_encmlp_destroy(hh)


def image_cases(lib):
    w = h = 2048
    yy, xx = np.meshgrid(np.linspace(0, 1, h, dtype=np.float32), np.linspace(0, 1, w, dtype=np.float32), indexing="ij")
    img = np.stack([0.5 + 0.5 * np.sin(40 * xx) * np.cos(31 * yy), xx * yy, 0.5 + 0.5 * np.sin(90 * (xx + yy)), np.ones_like(xx)], axis=-1).astype(np.float32)
    img = np.ascontiguousarray(img)
    for logb in (16, 18):
        cfg = A.image_encmlp_config(image_resolution=w)
        hh = C.c_void_p(); A.check(lib, lib.ngp_encmlp_create(C.byref(cfg), C.c_uint64(1337), C.byref(hh)))
        o = A.default_image_options(batch_size=1 << logb)
        t = C.c_void_p(); A.check(lib, lib.ngp_image_create(hh, ptr(img), A.IMAGE_FLOAT, w, h, C.byref(o), C.byref(t)))
        A.check(lib, lib.ngp_image_train(t, None, 100)); torch.cuda.synchronize()
        ms = timed(lambda: A.check(lib, lib.ngp_image_train(t, None, 50))) / 50
        loss = C.c_float(); A.check(lib, lib.ngp_image_loss(t, None, C.byref(loss)))
        print(json.dumps({"op": "image training step (pixel batch + fwd / bwd + optimizer)", "image": f"{w}x{h} float", "batch": 1 << logb, "ms": round(ms, 4), "samples_per_s": (1 << logb) / ms * 1e3, "loss": loss.value}), flush=True)
        lib.ngp_image_destroy(t); lib.ngp_encmlp_destroy(hh)


def main():
    lib = A.load_hip()
    which = sys.argv[1:] or ["sdf", "image"]
    if "sdf" in which:
        sdf_cases(lib)
    if "image" in which:
        image_cases(lib)


if __name__ == "__main__":
    main()
