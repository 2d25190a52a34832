#!/bin/bash
# round 5, call f: the sharded data-parallel step -- world-2 shared-GPU equivalence with the all-reduce step, the one-rank RCCL structure test, and what the structure costs at world 1
R=$PWD; O=gpurun_out/r05f; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_dist.py -q -x -s -m gpu -k "sharded or world1" > $O/pytest_dist.log 2>&1; grep -E "sharded|rccl world|passed|failed|Error|assert" $O/pytest_dist.log | cut -c1-500 | tail -20
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-f4-legs --no-calibration --eval-views 0"
for i in 1 2; do
  for v in plain dp_sharded dp_sharded_nosplit; do
    case $v in plain) E="NGP_X=1";; dp_allreduce) E="NGP_FORCE_DP=1";; dp_sharded) E="NGP_FORCE_DP=1 NGP_DP_SHARDED=1";; dp_sharded_nosplit) E="NGP_FORCE_DP=1 NGP_DP_SHARDED=1 NGP_DP_NO_SPLIT=1";; esac
    env $E timeout 200 $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
    print("$v $i", round(d['ms_per_step'],4), round(d['value']/1e6,2), d['config'].get('dp_backend'), 'loss', round(d['config']['loss'],7))
except Exception as e:
    print("$v $i FAILED", e, open("$O/bench_${v}_$i.err").read()[-600:])
PY
  done
done
