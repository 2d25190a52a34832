#!/bin/bash
# round 6 (second session), call bw: k1_count with max_mip + 1 levels in LDS (fox: 12 KiB -> five workgroups per CU) and a grid of 5 per CU (fox: <= 4 rays per workgroup = one batch each)
R=$PWD; O=gpurun_out/r06bw; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for pass in 1 2 3; do for v in "2 4" "1 5" "1 4" "2 5"; do set -- $v
  NGP_K1_LDS_EXTRA=$1 NGP_K1_MC_BLOCKS=$2 timeout 300 python bench.py --gpus 1 --scene fox --pretrain 3000 --steps 200 --warmup 20 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --no-calibration --eval-views 0 --profile-steps 32 > $O/fox_e$1_b$2_p$pass.json 2> $O/fox_e$1_b$2_p$pass.err || tail -3 $O/fox_e$1_b$2_p$pass.err
  python - $O/fox_e$1_b$2_p$pass.json "extra $1 blocks $2" $pass <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "pass", sys.argv[3], round(d["ms_per_step"] * 1000, 1), "us/step", {k: round(v * 1000, 1) for k, v in list(d["roofline"].get("kernel_ms_per_step", {}).items())[:3]})
PY
done; done
