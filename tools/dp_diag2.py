#!/usr/bin/env python
"""What does an RCCL communicator do to the kernels of its process?  (tools/dp_diag.py showed: +0.46 ms per training step as soon as a one-rank
communicator exists, whatever the step's structure, no RCCL call issued, host not the bottleneck.)  One process, three states -- before
ngp_comm_init, with the communicator, after ngp_comm_destroy -- and in each: (i) a loop of 2000 trivial torch kernels (per-launch cost),
(ii) one large streaming kernel (bandwidth), (iii) 200 fused training steps.  NCCL_DEBUG=INFO in the environment shows what RCCL sets up."""
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


def main():
    torch.cuda.set_device(0)
    lib = A.load_hip()

    class Args: pass
    args = Args(); args.scene = "synthetic"; args.images = 100; args.res = 800; args.eval_views = 0; args.eval_res = 400
    scene = bench.load_scene(args)
    cfg, opts, model, nerf = bench.make_trainer(lib, scene, 1 << 18)
    A.check(lib, lib.ngp_nerf_train(nerf, None, 1000))
    torch.cuda.synchronize()
    x = torch.zeros(1024, device="cuda")
    big = torch.zeros(1 << 28, device="cuda", dtype=torch.float32)  # 1 GiB
    out = {}

    def measure(tag):
        r = {}
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2000):
            x.add_(1.0)
        e1.record(); torch.cuda.synchronize()
        r["tiny_kernel_us_gpu"] = round(1e3 * e0.elapsed_time(e1) / 2000, 3)
        e0.record()
        for _ in range(10):
            big.add_(1.0)
        e1.record(); torch.cuda.synchronize()
        r["stream_1GiB_rw_GBps"] = round(10 * 2 * big.numel() * 4 / (1e6 * e0.elapsed_time(e1)), 1)
        A.check(lib, lib.ngp_nerf_train(nerf, None, 20)); torch.cuda.synchronize()
        t0 = time.perf_counter()
        A.check(lib, lib.ngp_nerf_train(nerf, None, 200))
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        r["train_ms_per_step"] = round(1e3 * (t2 - t0) / 200, 4); r["train_host_ms_per_step"] = round(1e3 * (t1 - t0) / 200, 4)
        r["threads"] = len(os.listdir("/proc/self/task"))
        out[tag] = r
        print(tag, r, file=sys.stderr, flush=True)

    measure("before_comm")
    os.environ["NGP_DP_FUSED_STEP"] = "1"  # read once by the library at the first ngp_nerf_train with a communicator... (static): keep the fused step throughout
    buf = (C.c_uint8 * 128)()
    A.check(lib, lib.ngp_comm_unique_id(buf))
    A.check(lib, lib.ngp_comm_init(nerf, 0, 1, buf))
    measure("with_comm")
    A.check(lib, lib.ngp_comm_destroy(nerf))
    measure("after_comm_destroy")
    print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith(("NGP_", "NCCL_", "RCCL_", "HSA_", "GPU_", "HIP_"))}, **out}))
    lib.ngp_nerf_destroy(nerf); lib.ngp_model_destroy(model)


if __name__ == "__main__":
    main()
