#!/bin/bash
# round 5, call b: k_train_fused (T1 + W in one kernel, packed fragment conversions) -- parity (bit-identical to the two kernels), then interleaved A / B with the driver's command
R=$PWD; O=gpurun_out/r05b; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -m gpu -k "t1_reuses or training_loop_tracks or lazy_k2 or fused_optimizer" > $O/pytest_train.log 2>&1; tail -8 $O/pytest_train.log | cut -c1-700
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "training_step" > $O/pytest_model.log 2>&1; tail -3 $O/pytest_model.log | cut -c1-400
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-f4-legs --no-calibration --eval-views 0"
for i in 1 2 3; do
  for v in fused two_kernels; do
    case $v in fused) E="NGP_X=1";; two_kernels) E="NGP_DEBUG_FLAGS2_OR=1";; esac
    env $E timeout 200 $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
k=d['roofline']['kernel_ms_per_step']
print("$v $i", round(d['ms_per_step'],4), round(d['value']/1e6,2), {n:round(v,4) for n,v in k.items() if 'train' in n or 'wgrad' in n}, 'loss', round(d['config']['loss'],7), d['roofline']['mfma'].get('k_train_fused'))
PY
  done
done
