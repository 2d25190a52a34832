#!/bin/bash
# round 4, call o: full GPU tier at HEAD + interleaved A/B of the new K1 with the driver's command
R=$PWD; O=gpurun_out/r04o; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0"
for i in 1 2 3; do
  for v in seg chunk; do
    case $v in seg) E="NGP_X=1";; chunk) E="NGP_DEBUG_FLAGS=33554432";; esac
    env $E timeout 200 $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
k=d['roofline']['kernel_ms_per_step']
print("$v $i", round(d['ms_per_step'],4), round(d['value']/1e6,2), {n:v for n,v in k.items() if 'generate' in n}, 'marched', d['config']['marched_samples_last_step'], 'loss', round(d['config']['loss'],7))
PY
  done
done
timeout 900 python -m pytest tests -q -x -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
