#!/usr/bin/env python
"""Operator micro-benchmarks of SURVEY.md 8(d) item 6 (GPU box): the fused encoding + MLP kernels in isolation on N = 2^18 and 2^22
positions, (i) i.i.d. uniform in the unit cube (worst-case locality: every hashed level misses), (ii) ray-coherent (32 consecutive
steps per ray).  Tables U(-1, 1) (trained-like magnitudes), upstream gradients N(0,1) * 128 / N in half.  One JSON line per case:
milliseconds (median of 5), samples/s and the ALGORITHMIC GB/s (548 B per forward sample, 1,572 B per training-step sample,
38 B per parameter) -- the same accounting as bench.py's roofline object.
usage: python tools/op_bench.py"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "instant-ngp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ngp_abi as A  # noqa: E402
from common import HipModel, dptr, ptr, random_coords  # noqa: E402


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return float(np.median(ms))


def main():
    lib = A.load_hip()
    cfg = A.base_model_config(1)
    hm = HipModel(lib, cfg)
    rng = np.random.default_rng(1)
    p = np.empty(hm.n, np.float32)
    p[:hm.n_mlp] = rng.uniform(-0.3, 0.3, hm.n_mlp); p[hm.n_mlp:] = rng.uniform(-1, 1, hm.n - hm.n_mlp)
    hm.set_params(p)
    for logn in (18, 22):
        n = 1 << logn
        for name, coh in (("iid", False), ("ray_coherent", True)):
            c = torch.from_numpy(random_coords(n, seed=logn, ray_coherent=coh)).cuda()
            out = torch.zeros((n, 4), dtype=torch.int16, device="cuda")
            ms = timed(lambda: A.check(lib, lib.ngp_model_inference(hm.h, None, dptr(c), 7, n, None, dptr(out), 4, 0)))
            print(json.dumps({"op": "nerf_inference (encoding + both MLPs)", "n": n, "positions": name, "ms": round(ms, 4), "samples_per_s": n / ms * 1e3, "algorithmic_GBps": 548 * n / ms / 1e6}))
            pos = c[:, :3].contiguous(); dout = torch.zeros((n,), dtype=torch.int16, device="cuda")
            ms = timed(lambda: A.check(lib, lib.ngp_model_density(hm.h, None, dptr(pos), 3, n, dptr(dout), 1, 0)))
            print(json.dumps({"op": "density (encoding + density MLP)", "n": n, "positions": name, "ms": round(ms, 4), "samples_per_s": n / ms * 1e3, "algorithmic_GBps": (12 + 512 + 2) * n / ms / 1e6}))
            if logn == 18:
                dl = torch.from_numpy((rng.normal(size=(n, 4)) * (128.0 / n)).astype(np.float16).view(np.int16)).cuda()
                for flags, label in ((0, "binned hashed levels"), (2048, "all levels through atomics")):
                    lib.ngp_debug_set_flags(flags | 4096)
                    ms = timed(lambda: A.check(lib, lib.ngp_model_training_step(hm.h, None, dptr(c), 7, n, dptr(dl), 4)))
                    lib.ngp_debug_set_flags(0)
                    print(json.dumps({"op": f"training_step (fwd + bwd + scatter + wgrad; {label})", "n": n, "positions": name, "ms": round(ms, 4), "samples_per_s": n / ms * 1e3, "algorithmic_GBps": 1572 * n / ms / 1e6}))
    ms = timed(lambda: A.check(lib, lib.ngp_model_optimizer_step(hm.h, None, C.c_float(128.0))))
    print(json.dumps({"op": "optimizer_step (Adam + EMA, gradients of the last training step)", "n_params": hm.n, "ms": round(ms, 4), "algorithmic_GBps": 38 * hm.n / ms / 1e6}))
    for which, cfg2 in (("image (2-D, all levels dense)", A.image_encmlp_config()), ("sdf (3-D, 5 dense + 11 hashed levels)", A.sdf_encmlp_config())):
        h = C.c_void_p()
        A.check(lib, lib.ngp_encmlp_create(C.byref(cfg2), C.c_uint64(1337), C.byref(h)))
        a, b = C.c_uint64(), C.c_uint64(); lib.ngp_encmlp_n_params(h, C.byref(a), C.byref(b))
        q = np.empty(a.value, np.float32); q[:b.value] = rng.uniform(-0.3, 0.3, b.value); q[b.value:] = rng.uniform(-1, 1, a.value - b.value)
        A.check(lib, lib.ngp_encmlp_set_params_host(h, ptr(q), C.c_uint64(a.value)))
        D, no = cfg2.n_pos_dims, cfg2.n_output_dims
        for logn in (16, 22):
            n = 1 << logn
            x = torch.rand((n, D), device="cuda"); o = torch.zeros((n, no), dtype=torch.int16, device="cuda")
            ms = timed(lambda: A.check(lib, lib.ngp_encmlp_inference(h, None, dptr(x), D, n, dptr(o), no)))
            bytes_per = 4 * D + 16 * (1 << D) * 4 + 2 * no
            print(json.dumps({"op": f"encmlp_inference {which}", "n": n, "positions": "iid", "ms": round(ms, 4), "samples_per_s": n / ms * 1e3, "algorithmic_GBps": bytes_per * n / ms / 1e6}))
        lib.ngp_encmlp_destroy(h)


if __name__ == "__main__":
    main()
