#!/usr/bin/env python
"""What tcnn's two EMA kernels are worth in test PSNR (round 6: the product switched from a hybrid to tcnn's default, the half-precision state): two trainings per seed that differ ONLY in
`ema_full_precision` -- run with the deterministic K3 compaction (DBG_K3_TWO_PASS), so that both see bit-identical batches and parameters and the PSNR difference isolates the inference
(EMA) weights the evaluation renders with.
usage (GPU box): python tools/ab_ema.py <scene: synthetic|hard|fox> <steps, e.g. 2000,10000> <n_seeds> > out.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ngp_abi as A  # noqa: E402
import numpy as np  # noqa: E402


def main():
    scene_name = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
    steps = sorted(int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2000,5000").split(","))
    n_seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 3

    class Args:
        scene = scene_name; images = 100; res = 800; eval_views = 8; eval_res = 400; eval_spp = 1; batch = 1 << 18
    lib = A.load_hip(); A.check(lib, lib.ngp_init())
    scene = bench.load_scene(Args, scene_name)
    out = {"scene": scene["name"], "steps": steps, "eval": f"{len(scene['eval'])} {scene['eval_kind']}", "deterministic_k3": True, "per_seed": {}}
    lib.ngp_debug_set_flags(1048576)  # DBG_K3_TWO_PASS
    for seed in range(1337, 1337 + n_seeds):
        row = {}
        for full in (0, 1):
            _, _, model, nerf = bench.make_trainer(lib, scene, Args.batch, seed=seed, model_kw={"ema_full_precision": full})
            done, ps = 0, []
            for k in steps:
                A.check(lib, lib.ngp_nerf_train(nerf, None, k - done)); done = k
                ps.append(round(bench.eval_psnr(lib, nerf, scene, Args.eval_spp), 4))
            st = bench.get_stats(lib, nerf)
            row["full_precision" if full else "half_precision"] = {"psnr_db": ps, "loss": float(st.loss)}
            lib.ngp_nerf_destroy(nerf); lib.ngp_model_destroy(model)
        row["delta_db_half_minus_full"] = [round(a - b, 4) for a, b in zip(row["half_precision"]["psnr_db"], row["full_precision"]["psnr_db"])]
        row["identical_training"] = row["half_precision"]["loss"] == row["full_precision"]["loss"]
        out["per_seed"][str(seed)] = row
    lib.ngp_debug_set_flags(0)
    d = np.array([r["delta_db_half_minus_full"] for r in out["per_seed"].values()])
    out["mean_delta_db"] = [round(float(x), 4) for x in d.mean(0)]
    out["max_abs_delta_db"] = round(float(np.abs(d).max()), 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
