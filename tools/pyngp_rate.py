#!/usr/bin/env python
"""Training rate through the OUTER boundary (pyngp.Testbed.frame(), the loop scripts/run.py runs) on the bench's lego-format stand-in (100 views 800 x 800, base.json, batch 2^18):
how far is `while testbed.frame()` from the C-ABI loop bench.py times?  usage: python tools/pyngp_rate.py [warm_steps] [timed_steps]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "instant-ngp_amd"), os.path.join(ROOT, "instant-ngp_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth_scene  # noqa: E402
import pyngp as ngp  # noqa: E402


def main():
    warm = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    timed = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    d = tempfile.mkdtemp(prefix="ngp_scene_")
    synth_scene.write_dataset(d, n_train=100, n_test=1, res=800, device="cuda")
    t = ngp.Testbed()
    t.load_training_data(os.path.join(d, "transforms_train.json"))
    t.reload_network_from_file("")
    t.shall_train = True
    for _ in range(warm):
        t.frame()
    _ = t.loss
    t0 = time.perf_counter()
    for _ in range(timed):
        t.frame()
    import torch
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"pyngp Testbed.frame(): {timed} steps in {dt:.3f} s = {dt / timed * 1e3:.4f} ms per step (loss read-back every 16 steps, as the reference); training_step {t.training_step}, loss {t.loss:.6f}")


if __name__ == "__main__":
    main()
