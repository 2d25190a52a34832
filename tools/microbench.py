#!/usr/bin/env python
"""Ablation micro-benchmark (GPU box): per-kernel HIP-event timings of the training step under debug switches."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "instant-ngp_amd"), os.path.join(ROOT, "tests")]
import torch
import ngp_abi as A
import synth_scene


def main():
    pretrain = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    lib = A.load_hip()
    images, xforms, meta, _ = synth_scene.make_dataset(100, 800, "cuda")
    n = len(images)
    M = (A.ImageMeta * n)(); X = (A.Xform * n)()
    for i in range(n):
        M[i].pixels = images[i].data_ptr(); M[i].image_data_type = A.IMAGE_BYTE; M[i].lens_mode = 0
        M[i].resolution[0] = M[i].resolution[1] = 800; M[i].principal_point[0] = M[i].principal_point[1] = 0.5
        M[i].focal_length[0], M[i].focal_length[1] = meta["focal_length"]
        for k in range(12):
            X[i].start[k] = X[i].end[k] = float(xforms[i][k])
    cfg = A.base_model_config(1)
    model = C.c_void_p(); A.check(lib, lib.ngp_model_create(C.byref(cfg), C.c_uint64(1337), C.byref(model)))
    opts = A.default_nerf_options(1)
    nerf = C.c_void_p(); A.check(lib, lib.ngp_nerf_create(model, C.byref(opts), A.scene_aabb(1), C.byref(nerf)))
    A.check(lib, lib.ngp_nerf_set_dataset_device(nerf, n, M, X))
    A.check(lib, lib.ngp_nerf_train(nerf, None, pretrain))
    lib.ngp_profile_name.restype = C.c_char_p
    npf = lib.ngp_profile_count()
    variants = [("t1_no_binning", 2048), ("default", 0), ("t1_occ2", 1024), ("default_again", 0), ("t1_occ2_again", 1024), ("fwd_pair_loads", 256), ("fwd_occ4", 512), ("t1_no_quads", 128), ("t1_no_pair_halves", 64), ("k3_thread_per_ray", 32), ("t1_no_merge", 16), ("t1_no_scatter", 2)]
    if len(sys.argv) > 3:
        keep = sys.argv[3].split(",")
        variants = [v for v in variants if v[0] in keep]
    res = {}
    for name, flags in variants:
        lib.ngp_debug_set_flags(flags)
        A.check(lib, lib.ngp_nerf_train(nerf, None, 4))
        lib.ngp_profile_enable(1)
        A.check(lib, lib.ngp_nerf_train(nerf, None, nsteps))
        ms = (C.c_double * npf)(); cnt = (C.c_uint64 * npf)()
        lib.ngp_profile_read(ms, cnt); lib.ngp_profile_enable(0)
        s = A.NerfStats(); lib.ngp_nerf_get_stats(nerf, None, C.byref(s))
        res[name] = {lib.ngp_profile_name(i).decode(): round(ms[i] / nsteps, 4) for i in range(npf) if cnt[i]}
        res[name]["_rays_per_batch"] = s.rays_per_batch; res[name]["_before"] = s.measured_batch_size_before_compaction; res[name]["_loss"] = s.loss
        print(name, json.dumps(res[name]), flush=True)
    lib.ngp_debug_set_flags(0)


if __name__ == "__main__":
    main()
