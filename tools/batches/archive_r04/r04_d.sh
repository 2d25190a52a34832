#!/bin/bash
# round 4, call d: how much of K1 hides under K2 / K3 (dummy K1 on the side stream), fused-optimizer equivalence test
R=$PWD; O=gpurun_out/r04d; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -m pytest tests/test_gpu_train.py -q -s -k "fused_optimizer" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0"
for i in 1 2; do
  for v in base dummy dummy_k2b2 base_k2b2; do
    case $v in base) E="NGP_X=1";; dummy) E="NGP_EXP_DUMMY_K1=1";; dummy_k2b2) E="NGP_EXP_DUMMY_K1=1 NGP_K2_BLOCKS_PER_CU=2";; base_k2b2) E="NGP_K2_BLOCKS_PER_CU=2";; esac
    env $E $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
k=d['roofline']['kernel_ms_per_step']
print("$v $i", round(d['ms_per_step'],4), round(d['value']/1e6,2), {a:k[a] for a in list(k)[:5]})
PY
  done
done
