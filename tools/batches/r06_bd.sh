#!/bin/bash
# round 6 (second session), call bd: per-kernel HIP-event times of the fox leg (BASELINE.json configs[2]: aabb_scale 4, real capture) and of the hard stand-in
R=$PWD; O=gpurun_out/r06bd; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for sc in fox hard; do
timeout 300 python bench.py --gpus 1 --scene $sc --pretrain 5000 --steps 200 --warmup 20 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --no-calibration --eval-views 0 --profile-steps 32 > $O/$sc.json 2> $O/$sc.err || tail -3 $O/$sc.err
python - $O/$sc.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"][:80], round(d["ms_per_step"] * 1000, 1), "us/step", {k: round(v * 1000, 1) for k, v in d["roofline"].get("kernel_ms_per_step", {}).items()})
print({k: d["config"].get(k) for k in ("rays_per_step", "samples_per_ray_compacted", "marched_samples_per_hit_ray")})
PY
done
