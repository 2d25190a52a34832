#!/bin/bash
# round 5, call d: timing experiments on k_train_fused (NGP_FUSED_DBG: 1 = no dependent src_index load, 2 = cache-resident encodings, 3 = prologue only)
R=$PWD; O=gpurun_out/r05d; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
B="python bench.py --gpus 1 --steps 20 --warmup 5 --pretrain 300 --no-cpu-baseline --no-fox-leg --no-f4-legs --no-calibration --eval-views 0"
for i in 1 2; do
  for v in 0 1 2 3; do
    NGP_FUSED_DBG=$v timeout 200 $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
print("dbg $v run $i", round(d['ms_per_step'],4), d['roofline']['mfma'].get('k_train_fused',{}).get('avg_launch_ms'))
PY
  done
done
