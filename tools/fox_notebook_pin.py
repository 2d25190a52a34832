#!/usr/bin/env python
"""The one number in the mount that a real instant-ngp build produced: /root/reference/notebooks/instant_ngp.ipynb (cell 23) ran
`scripts/run.py <fox> --n_steps 2000` with the 2022 configs/nerf/base.json (`GridEncoding: Nmin=16 b=1.51572 F=2 T=2^19 L=16`, density MLP 64 x 1 hidden,
colour MLP 64 x 2 hidden, full precision CutlassMLP on a T4) and its progress bar ended with `loss=0.00102` -- Testbed.loss = the loss of the last step whose
loss was read back (every 16th, testbed.cu:4625), a single-batch value.  This script trains data/nerf/fox through pyngp with configs/nerf/base_l16f2.json for the
same number of steps and prints the same quantity together with its spread over the last steps, so that the comparison carries its own noise bar.
usage: python tools/fox_notebook_pin.py [n_steps] [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instant-ngp_amd"))
import numpy as np

NOTEBOOK_LOSS, NOTEBOOK_STEPS = 0.00102, 2000


def run(n_steps=NOTEBOOK_STEPS, config="base_l16f2.json", seed=1337):
    import pyngp as ngp
    scene = os.path.join(ROOT, "_ref_data", "data", "nerf", "fox", "transforms.json")
    if not os.path.exists(scene):
        raise RuntimeError("_ref_data/data/nerf/fox is not staged (tools/stage_reference_data.py)")
    t = ngp.Testbed()
    t.seed = seed
    t.load_training_data(scene)
    t.reload_network_from_file(os.path.join(ROOT, "instant-ngp_amd", "configs", "nerf", config))
    t.shall_train = True
    losses = []
    t0 = time.perf_counter()
    while t.training_step < n_steps:
        t.train(1 << 18)
        if t.training_step % 16 == 0:
            losses.append((int(t.training_step), float(t.loss)))
    wall = time.perf_counter() - t0
    tail = np.array([l for s, l in losses if s > n_steps - 320])  # the last 20 read-backs
    return {"scene": "data/nerf/fox (50 JPEGs 1080x1920, aabb_scale 4)", "config": config, "n_steps": n_steps, "seed": seed, "n_images": int(t.nerf.training.dataset.n_images),
            "loss_last_readback": losses[-1][1], "loss_last_readback_step": losses[-1][0],
            "loss_tail_mean": float(tail.mean()), "loss_tail_std": float(tail.std(ddof=1)), "loss_tail_min": float(tail.min()), "loss_tail_max": float(tail.max()), "loss_tail_n": int(tail.size),
            "notebook_loss": NOTEBOOK_LOSS, "notebook_steps": NOTEBOOK_STEPS, "ratio_tail_mean_to_notebook": float(tail.mean() / NOTEBOOK_LOSS),
            "loss_curve_every_160": [(s, round(l, 6)) for s, l in losses if s % 160 == 0], "wall_s": round(wall, 2)}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else NOTEBOOK_STEPS
    r = {"l16f2": run(n, "base_l16f2.json"), "l8f4_current_base_json": run(n, "base.json")}
    s = json.dumps(r)
    print(s)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(s)
