#!/bin/bash
# round 4, call h: table gathers with the nt (L1-bypassing) policy vs plain, interleaved
R=$PWD; O=gpurun_out/r04h; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
L=instant-ngp_amd/libngp_hip.so
cp $L /tmp/base.so; cp instant-ngp_amd/libngp_hip_exp_nt.so /tmp/nt.so
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0"
for i in 1 2 3; do
  for v in base nt; do
    cp /tmp/$v.so $L
    $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
k=d['roofline']['kernel_ms_per_step']
print("$v $i", round(d['ms_per_step'],4), round(d['value']/1e6,2), 'k2', k.get('k_inference'), 'density', k.get('k_inference<density_only>'), 'loss', d['config']['loss'])
PY
  done
done
cp /tmp/base.so $L
