#!/usr/bin/env python
"""Evaluation-render timing (GPU box): train the synthetic lego-format scene, then time the run.py PSNR procedure
(8 held-out views 800x800, spp 8, EMA weights).  usage: render_bench.py [train_steps] [views] [res] [spp]"""
import sys
import time
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-ngp_amd"), os.path.join(ROOT, "tests")]
import argparse
import torch
import bench
import ngp_abi as A


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    views = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    res = int(sys.argv[3]) if len(sys.argv) > 3 else 800
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    args = argparse.Namespace(scene="synthetic", images=100, res=800, eval_views=views, eval_res=res, batch=1 << 18)
    lib = A.load_hip()
    scene = bench.load_scene(args)
    cfg, opts, model, nerf = bench.make_trainer(lib, scene, args.batch)
    A.check(lib, lib.ngp_nerf_train(nerf, None, steps))
    torch.cuda.synchronize()
    bench.eval_psnr(lib, nerf, scene, 1)  # warm-up (buffer allocation)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    psnr = bench.eval_psnr(lib, nerf, scene, spp)
    dt = time.perf_counter() - t0
    print(f"eval {views} views {res}x{res} spp {spp}: {dt:.3f} s ({dt / (views * spp) * 1e3:.2f} ms per frame), PSNR {psnr:.3f} dB after {steps} steps")


if __name__ == "__main__":
    main()
