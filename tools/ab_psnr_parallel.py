#!/usr/bin/env python
"""Equal-step PSNR of the production path vs the reference-order path over MANY seeds: `bench.py --ab-only` once per seed range, the ranges side by
side on the one GPU -- the reference-order path (thread-per-ray K1, sequential K3, half atomics) keeps less than one wavefront per SIMD busy, so several trainings overlap
almost for free -- and the per-seed results merged: per path mean / std, the PAIRED difference production - reference_order with its standard error and the 95 % interval
(Student t), and whether that interval lies inside +- 0.1 dB (the north star's tolerance).
usage: ab_psnr_parallel.py <out.json> <scene: synthetic|fox> <steps, e.g. 2000,5000> <n_seeds> <n_workers> [extra bench.py args...]   (AB_SEED0: first seed, default 1337)
       ab_psnr_parallel.py --merge <out.json> <in1.json> <in2.json> ...      (pool the seeds of several runs of the same experiment)"""
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T95 = {2: 12.706, 3: 4.303, 4: 3.182, 5: 2.776, 6: 2.571, 7: 2.447, 8: 2.365, 9: 2.306, 10: 2.262, 11: 2.228, 12: 2.201, 13: 2.179, 14: 2.160, 15: 2.145, 16: 2.131, 20: 2.093, 24: 2.069}


def summarise(merged, keys):
    n = len(merged["seeds"])
    t = T95.get(n, 2.0 if n > 24 else T95[max(k for k in T95 if k <= n)])
    for name in ("production", "reference_order"):
        ps = merged[name + "_per_seed"]
        merged[name + "_mean_db"] = {k: round(sum(v) / len(v), 4) for k, v in ps.items()}
        merged[name + "_std_db"] = {k: round(math.sqrt(sum((x - sum(v) / len(v)) ** 2 for x in v) / (len(v) - 1)), 4) for k, v in ps.items()}
    merged["paired"] = {}
    ok = True
    for k in keys:
        d = [a - b for a, b in zip(merged["production_per_seed"][k], merged["reference_order_per_seed"][k])]
        mean = sum(d) / n
        sd = math.sqrt(sum((x - mean) ** 2 for x in d) / (n - 1))
        half = t * sd / math.sqrt(n)
        inside = abs(mean) + half <= 0.1
        ok = ok and inside
        merged["paired"][k] = {"delta_db_per_seed": [round(x, 4) for x in d], "mean_db": round(mean, 4), "std_db": round(sd, 4), "stderr_db": round(sd / math.sqrt(n), 4),
                               "ci95_db": [round(mean - half, 4), round(mean + half, 4)], "ci95_inside_0p1_db": inside}
    merged["within_0p1_db"] = ok
    return ok


def merge_files(out, files):
    parts = [json.load(open(f)) for f in files]
    keys = list(parts[0]["production_per_seed"].keys())
    merged = {k: parts[0][k] for k in ("what", "scene", "eval", "steps", "reference_order_flags", "extra_args") if k in parts[0]}
    merged["seeds"] = sum((p["seeds"] for p in parts), []); merged["pooled_from"] = [os.path.basename(f) for f in files]; merged["wall_seconds"] = round(sum(p.get("wall_seconds", 0) for p in parts), 1)
    assert len(set(merged["seeds"])) == len(merged["seeds"]), "the runs share seeds"
    for name in ("production", "reference_order"):
        merged[name + "_per_seed"] = {k: sum((p[name + "_per_seed"][k] for p in parts), []) for k in keys}
    ok = summarise(merged, keys)
    json.dump(merged, open(out, "w"), indent=1)
    print(json.dumps({k: {q: merged["paired"][k][q] for q in ("mean_db", "stderr_db", "ci95_db", "ci95_inside_0p1_db")} for k in keys}), "seeds", len(merged["seeds"]), "within_0p1_db", ok)


def main():
    if sys.argv[1] == "--merge":
        return merge_files(sys.argv[2], sys.argv[3:])
    out, scene, steps, n_seeds, n_workers = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    extra = sys.argv[6:]
    per = (n_seeds + n_workers - 1) // n_workers
    procs = []
    t0 = time.time()
    for w in range(n_workers):
        s0, cnt = int(os.environ.get("AB_SEED0", "1337")) + w * per, min(per, n_seeds - w * per)
        if cnt <= 0:
            break
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--ab-only", "--scene", scene, "--ab-psnr", steps, "--ab-seeds", str(cnt), "--ab-seed0", str(s0), "--no-calibration"] + extra
        procs.append((s0, cnt, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    parts = []
    for s0, cnt, p in procs:
        o, e = p.communicate()
        lines = [l for l in o.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not lines:
            print(f"worker seeds {s0}+{cnt} failed:\n{e[-2000:]}")
            continue
        parts.append(json.loads(lines[-1]))
    wall = time.time() - t0
    assert parts, "no worker finished"
    keys = list(parts[0]["production_per_seed"].keys())
    merged = {"what": "tools/ab_psnr_parallel.py: production - reference_order PSNR at equal step counts, paired by seed", "scene": parts[0]["scene"], "eval": parts[0]["eval"], "steps": parts[0]["steps"],
              "reference_order_flags": parts[0]["reference_order_flags"], "seeds": sum((p["seeds"] for p in parts), []), "workers": len(parts), "wall_seconds": round(wall, 1), "extra_args": extra}
    for name in ("production", "reference_order"):
        merged[name + "_per_seed"] = {k: sum((p[name + "_per_seed"][k] for p in parts), []) for k in keys}
    ok = summarise(merged, keys)
    json.dump(merged, open(out, "w"), indent=1)
    print(json.dumps({k: merged["paired"][k] for k in keys}), "wall", round(wall, 1), "s", "within_0p1_db", ok)


if __name__ == "__main__":
    main()
