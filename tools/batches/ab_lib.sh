#!/bin/bash
# helpers for interleaved A/B runs of the headline workload on one box (source this from a batch script)
#   ab_run <tag> [ENV=value ...]   400 timed steps + 32 per-kernel profile steps; prints ms per step and the per-kernel HIP-event times (us)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
AB_BENCH="python bench.py --gpus 1 --steps ${AB_STEPS:-400} --warmup 20 --no-cpu-baseline --no-fox-leg --no-f4-legs --no-calibration --eval-views 0 --profile-steps ${AB_PROFILE_STEPS:-32}"
ab_run() {
  local tag=$1; shift
  env "$@" timeout 180 $AB_BENCH > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; tail -3 $O/$tag.err; return; }
  python - "$O/$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["roofline"].get("kernel_ms_per_step", {})
short = {"k_train_fwd_bwd+k_grad_bin+k_grad_accumulate": "scatter", "k_inference": "K2", "k_generate_training_samples": "K1", "k_compute_loss": "K3", "k_inference<density_only>": "gridinf",
         "occupancy_grid_misc": "gridmisc", "k_optimizer": "opt", "k_fill_rollover": "K4", "k_wgrad_reduce": "wred"}
print(f"{sys.argv[2]:28s} {d['ms_per_step']*1000:7.1f} us/step  {d['value']/1e6:6.1f} Mrays/s | " + " ".join(f"{short.get(n, n)} {v*1000:.1f}" for n, v in k.items()))
PY
}
