#!/usr/bin/env python
"""Ablation micro-benchmark (GPU box): per-kernel HIP-event timings of the training step under debug switches and layout knobs.
usage: microbench.py [pretrain] [steps] [comma separated variant names]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "instant-ngp_amd"), os.path.join(ROOT, "tests")]
import torch
import ngp_abi as A
import synth_scene

# name: (debug flags, (bin chunk log2, split, cap), (k2 rounds, k2 tile width))
VARIANTS = [
    ("default", 0, (11, 0, 0), (1, 16)),   # 2048-entry chunks: the library default since round 6
    ("bin_no_dense_merge", 16777216, (12, 0, 0), (1, 16)),   # dense levels in the bin lists without run merging
    ("t1_dense_atomics", 8388608, (12, 0, 0), (1, 16)),   # dense levels as half atomics from T1 instead of through the bin lists
    ("grid_no_sort", 4194304, (12, 0, 0), (1, 16)),   # occupancy-grid update in generation order
    ("k1_chunk_march", 33554432, (12, 0, 0), (1, 16)),     # chunk kernels k1_count / k1_write instead of the segment prepass + sample lists
    ("k1_no_prefilter", 2097152, (12, 0, 0), (1, 16)),   # k1_count without the coarse-occupancy prefilter in LDS
    ("k2_tile8", 0, (12, 0, 0), (1, 8)),
    ("k2_tile32", 0, (12, 0, 0), (1, 32)),
    ("k2_rounds3", 0, (12, 0, 0), (3, 32)),
    ("t1_dense_external", 262144 | 8388608, (12, 0, 0), (1, 16)),   # dense levels' atomics issued by k_grad_dense on its own stream instead of by T1
    ("w_single_role", 32768, (12, 0, 0), (1, 16)),       # round-1 weight-gradient kernel
    ("bin_no_hashed_merge", 65536, (12, 0, 0), (1, 16)), # k_grad_bin merges same-cell runs on the dense levels only
    ("separate_grad_memset", 131072, (12, 0, 0), (1, 16)),
    ("round1_backward", 32768 | 131072, (12, 1, 0), (1, 16)),
    ("bin_chunk12_split", 0, (12, 1, 0), (1, 16)),   # round-1 layout
    ("bin_chunk12", 0, (12, 0, 0), (1, 16)),   # rounds 2-5: 4096-entry chunks, one 128 KiB accumulate block per CU
    ("k2_tile16_r4", 0, (12, 0, 0), (4, 16)),
    ("k2_eager", 8192, (12, 0, 0), (1, 16)),
    ("k1_no_first_point_skip", 268435456, (12, 0, 0), (1, 16)),   # k1_count evaluates the chunks behind the ray's exit as well (rounds 1-2)
    ("k3_one_ray_per_wave", 134217728, (12, 0, 0), (1, 16)),   # K3 with a whole wavefront per ray (rounds 1-2)
    ("k1_independent_lattice", 16384, (12, 0, 0), (1, 16)),
    ("default_again", 0, (11, 0, 0), (1, 16)),
    ("t1_no_binning", 2048, (12, 0, 0), (1, 16)),
    ("t1_no_scatter", 2, (12, 0, 0), (1, 16)),
]


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
    variants = VARIANTS
    if len(sys.argv) > 3:
        keep = sys.argv[3].split(",")
        variants = [v for v in variants if v[0] in keep]
    for name, flags, (cl2, split, cap), (rounds, tw) in variants:
        lib.ngp_debug_set_flags(flags)
        A.check(lib, lib.ngp_debug_set_bin_params(cl2, split, cap))
        A.check(lib, lib.ngp_nerf_set_k2_params(nerf, rounds, tw))
        A.check(lib, lib.ngp_nerf_train(nerf, None, 4))
        # (a) un-profiled wall time per step (helper streams on), (b) per-kernel HIP-event times (everything on one stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        t0 = time.perf_counter()
        A.check(lib, lib.ngp_nerf_train(nerf, None, nsteps))
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / nsteps * 1e3
        lib.ngp_profile_enable(1)
        A.check(lib, lib.ngp_nerf_train(nerf, None, nsteps))
        ms = (C.c_double * npf)(); cnt = (C.c_uint64 * npf)()
        lib.ngp_profile_read(ms, cnt); lib.ngp_profile_enable(0)
        s = A.NerfStats(); lib.ngp_nerf_get_stats(nerf, None, C.byref(s))
        r = {lib.ngp_profile_name(i).decode(): round(ms[i] / nsteps, 4) for i in range(npf) if cnt[i]}
        r["_sum"] = round(sum(r.values()), 4); r["_wall_ms_per_step"] = round(wall, 4)
        r["_rays_per_batch"] = s.rays_per_batch; r["_rays_hit"] = s.n_rays_last; r["_before"] = s.measured_batch_size_before_compaction
        r["_k2_evals"] = s.network_evaluations; r["_loss"] = s.loss
        print(name, json.dumps(r), flush=True)
    lib.ngp_debug_set_flags(0); lib.ngp_debug_set_bin_params(11, 0, 0)


if __name__ == "__main__":
    main()
