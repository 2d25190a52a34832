#!/bin/bash
# round 5, call q: k1_count_segments with the next ray's record prefetched -- bit-exact K1 tests, then the driver's command against the previous commit's library, interleaved
R=$PWD; O=gpurun_out/r05q; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_nerf.py -q -x -m gpu -k "k1_" > $O/pytest_k1.log 2>&1; tail -3 $O/pytest_k1.log | cut -c1-600
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-f4-legs --no-calibration --eval-views 0"
for i in 1 2 3; do
  for v in new prev; do
    case $v in new) E="NGP_X=1";; prev) E="NGP_HIP_LIB=$R/instant-ngp_amd/ab/libngp_hip_prev.so";; esac
    env $E timeout 150 $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
k=d['roofline']['kernel_ms_per_step']
print("$v $i", round(d['ms_per_step'],4), round(d['value']/1e6,2), {a:round(b*1000,1) for a,b in k.items()}, 'loss', round(d['config']['loss'],7))
PY
  done
done
