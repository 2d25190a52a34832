#!/bin/bash
# round 5, call e: k_train_fused with the roles rebalanced (r1 tiles in role B) and shift-or ReLU bits vs the first fused version (ab/libngp_hip_prev.so) vs the two kernels; parity first
R=$PWD; O=gpurun_out/r05e; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -m gpu -k "t1_reuses or training_loop_tracks" > $O/pytest_train.log 2>&1; tail -5 $O/pytest_train.log | cut -c1-700
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-f4-legs --no-calibration --eval-views 0"
for i in 1 2 3; do
  for v in new prev two_kernels; do
    case $v in new) E="NGP_X=1";; prev) E="NGP_HIP_LIB=$R/instant-ngp_amd/ab/libngp_hip_prev.so";; two_kernels) E="NGP_DEBUG_FLAGS2_OR=1";; esac
    env $E timeout 200 $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
k=d['roofline']['kernel_ms_per_step']
print("$v $i", round(d['ms_per_step'],4), round(d['value']/1e6,2), 'fused', (d['roofline']['mfma'].get('k_train_fused') or {}).get('avg_launch_ms'), 'loss', round(d['config']['loss'],7))
PY
  done
done
