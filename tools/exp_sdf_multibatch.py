#!/usr/bin/env python
"""Experiment (round 6): does the SDF ground-truth launch last as long as its longest dependent chain?  If so, the ground truth of G training batches in ONE launch costs
little more than that of one.  Times ngp_sdf_signed_distance over the BVH halves (near-surface 3/8 + uniform 1/8) of 1, 2, 3, 4 consecutive training batches of armadillo.
usage: python tools/exp_sdf_multibatch.py   (NGP_SDF_WALK_OCC=2|4 selects workgroups per CU)"""
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


def timed(fn, reps=9):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return float(np.median(ms)), float(np.min(ms))


def main():
    import pyngp
    lib = A.load_hip()
    tris = np.ascontiguousarray(pyngp.read_obj(os.path.join(ROOT, "_ref_data", "data", "sdf", "armadillo.obj")))
    verts = tris.reshape(-1, 3).copy()
    box = A.Aabb(); scale = C.c_float()
    A.check(lib, lib.ngp_sdf_normalize_mesh_host(ptr(verts), C.c_uint64(len(verts)), C.byref(box), C.byref(scale)))
    tn = np.ascontiguousarray(verts.reshape(-1, 3, 3))
    cfg = A.sdf_encmlp_config()
    hh = C.c_void_p(); A.check(lib, lib.ngp_encmlp_create(C.byref(cfg), C.c_uint64(1337), C.byref(hh)))
    o = A.default_sdf_options()
    t = C.c_void_p(); A.check(lib, lib.ngp_sdf_create(hh, ptr(tn), len(tn), box, C.byref(o), C.byref(t)))
    B = int(o.batch_size); n_exact = B // 8 * 4; m = B - n_exact
    hip = C.CDLL("libamdhip64.so")
    batches = []
    for _ in range(4):
        A.check(lib, lib.ngp_sdf_train(t, None, 1)); torch.cuda.synchronize()
        pp, dp = C.c_void_p(), C.c_void_p(); lib.ngp_sdf_batch_ptrs(t, C.byref(pp), C.byref(dp))
        pos = torch.empty(B * 3, dtype=torch.float32, device="cuda")
        assert hip.hipMemcpy(C.c_void_p(pos.data_ptr()), pp, C.c_size_t(B * 12), 3) == 0  # device to device
        batches.append(pos)
    for G in (1, 2, 3, 4, 1):
        # order: near-surface points of every batch, then the uniform points of every batch (the walker hands distance items out from the END: long walks first)
        near = [b.view(-1, 3)[n_exact:B // 8 * 7] for b in batches[:G]]
        uni = [b.view(-1, 3)[B // 8 * 7:] for b in batches[:G]]
        pos = torch.cat(near + uni).contiguous()
        out = torch.zeros(pos.shape[0], dtype=torch.float32, device="cuda")
        med, mn = timed(lambda: A.check(lib, lib.ngp_sdf_signed_distance(t, None, C.c_void_p(pos.data_ptr()), pos.shape[0], C.c_void_p(out.data_ptr()))))
        print(json.dumps({"batches_per_launch": G, "points": pos.shape[0], "ms_median": round(med, 4), "ms_min": round(mn, 4), "ms_per_batch": round(med / G, 4),
                          "inside_fraction": round(float((out < 0).float().mean().item()), 4), "occ": os.environ.get("NGP_SDF_WALK_OCC", "2")}), flush=True)
    lib.ngp_sdf_destroy(t); lib.ngp_encmlp_destroy(hh)


if __name__ == "__main__":
    main()
