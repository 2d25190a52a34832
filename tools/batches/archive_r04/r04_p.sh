#!/bin/bash
# round 4, call p: extra dims (new kernels + trainer plumbing) on the GPU; regression of the stride-generic K1 / K3; A/B against the previous build of the library
R=$PWD; O=gpurun_out/r04p; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_extra_dims.py -q -x -s -m gpu > $O/pytest_extra.log 2>&1; tail -25 $O/pytest_extra.log | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_nerf.py -q -x -m gpu -k "k3_loss" > $O/pytest_regress.log 2>&1; tail -2 $O/pytest_regress.log
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0"
for i in 1 2 3; do
  for v in new head; do
    case $v in new) E="NGP_X=1";; head) E="NGP_HIP_LIB=$R/instant-ngp_amd/ab/libngp_hip_head.so";; esac
    env $E timeout 200 $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
k=d['roofline']['kernel_ms_per_step']
print("$v $i", round(d['ms_per_step'],4), round(d['value']/1e6,2), {n:v for n,v in k.items() if 'generate' in n or 'loss' in n}, 'loss', round(d['config']['loss'],7))
PY
  done
done
