#!/usr/bin/env python
"""Why does the data-parallel step cost 2x at world size 1?  One configuration per process (the library reads its diagnostics switches once):
    python tools/dp_diag.py <name> [comm] [pretrain] [steps]
prints one JSON line: wall ms per step, HOST enqueue ms per step (time until ngp_nerf_train returns, before the synchronize), threads of the process and
the CPU seconds each of them burnt during the timed region (a spinning proxy thread shows up here), cores available to the process.
Environment (csrc/ngp_api.hip): NGP_TRAIN_SPLIT_PHASES=1, NGP_DP_FUSED_STEP=1, NGP_DP_SKIP_ALLREDUCE=1."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "instant-ngp_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import torch
import ngp_abi as A
import bench


def thread_cpu():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (name, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
        except Exception:
            pass
    return out


def main():
    name = sys.argv[1]
    comm = len(sys.argv) > 2 and sys.argv[2] == "comm"
    pretrain = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
    torch.cuda.set_device(0)
    lib = A.load_hip()
    if os.environ.get("NGP_DEBUG_FLAGS"):
        lib.ngp_debug_set_flags(int(os.environ["NGP_DEBUG_FLAGS"], 0))

    class Args: pass
    args = Args(); args.scene = "synthetic"; args.images = 100; args.res = 800; args.eval_views = 0; args.eval_res = 400
    scene = bench.load_scene(args)
    cfg, opts, model, nerf = bench.make_trainer(lib, scene, 1 << 18)
    n_threads0 = len(os.listdir("/proc/self/task"))
    if comm:
        buf = (C.c_uint8 * 128)()
        A.check(lib, lib.ngp_comm_unique_id(buf))
        A.check(lib, lib.ngp_comm_init(nerf, 0, 1, buf))
    A.check(lib, lib.ngp_nerf_train(nerf, None, pretrain))
    torch.cuda.synchronize()
    A.check(lib, lib.ngp_nerf_train(nerf, None, 20))
    torch.cuda.synchronize()
    c0 = thread_cpu()
    s0 = bench.get_stats(lib, nerf)
    t0 = time.perf_counter()
    A.check(lib, lib.ngp_nerf_train(nerf, None, steps))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    c1 = thread_cpu()
    s1 = bench.get_stats(lib, nerf)
    busy = sorted(((c1[t][1] - c0.get(t, (None, 0.0))[1], c1[t][0], t) for t in c1), reverse=True)[:6]
    print(json.dumps({"name": name, "comm": comm, "env": {k: v for k, v in os.environ.items() if k.startswith("NGP_")},
                      "ms_per_step": round(1e3 * (t2 - t0) / steps, 4), "host_enqueue_ms_per_step": round(1e3 * (t1 - t0) / steps, 4),
                      "rays_per_s_M": round((s1.total_rays - s0.total_rays) / (t2 - t0) / 1e6, 2),
                      "threads_before_comm": n_threads0, "threads": len(c1), "cores_available": len(os.sched_getaffinity(0)),
                      "thread_cpu_s_in_timed_region": [(round(b, 3), n) for b, n, _ in busy], "timed_region_s": round(t2 - t0, 3)}))
    lib.ngp_nerf_destroy(nerf); lib.ngp_model_destroy(model)


if __name__ == "__main__":
    main()
