#!/bin/bash
# round 6 (second session), call bl: k_grad_accumulate with the listed levels dispatched last to first (NGP_ACC_REVERSE=1: the 1280 heavy blocks of the hashed levels start first, the
# dense levels' light blocks fill the half-empty third round) -- headline and fox, interleaved
R=$PWD; O=gpurun_out/r06bl; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
. tools/batches/ab_lib.sh
for pass in 1 2 3; do
  ab_run fwd_p$pass NGP_ACC_REVERSE=0
  ab_run rev_p$pass NGP_ACC_REVERSE=1
done
for pass in 1 2; do for v in 0 1; do
  NGP_ACC_REVERSE=$v timeout 300 python bench.py --gpus 1 --scene fox --pretrain 3000 --steps 200 --warmup 20 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --no-calibration --eval-views 0 --profile-steps 32 > $O/fox_rev${v}_p$pass.json 2> $O/fox_rev${v}_p$pass.err || tail -3 $O/fox_rev${v}_p$pass.err
  python - $O/fox_rev${v}_p$pass.json $v $pass <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("fox reverse", sys.argv[2], "pass", sys.argv[3], round(d["ms_per_step"] * 1000, 1), "us/step", {k: round(v * 1000, 1) for k, v in list(d["roofline"].get("kernel_ms_per_step", {}).items())[:3]})
PY
done; done
