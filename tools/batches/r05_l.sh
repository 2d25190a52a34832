#!/bin/bash
# round 5, call l: parity of the pcg32 skip-ahead tables (every K1 / K3 / grid-sampler thread), the two-load linear bitfield and the batched grid-sample candidates
# (bit-exact K1 / K3 / occupancy-grid tests); what a gather costs K2, its line or its lane address (tools/k2_lane_address.hip, timings only); then the driver's command,
# this build against the previous commit's library (instant-ngp_amd/ab/libngp_hip_prev.so), interleaved
R=$PWD; O=gpurun_out/r05l; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 500 python -m pytest tests/test_gpu_nerf.py -q -x -m gpu -k "occupancy_grid_chain or k1_sample_parallel or k1_sequential or k3_loss_and_compaction or fill_rollover or k1_sample_cap" > $O/pytest_nerf.log 2>&1; tail -5 $O/pytest_nerf.log | cut -c1-600
( cd /tmp; timeout 40 $R/tools/k2_lane_address 380000 20 > $R/$O/k2_lane_address.txt 2>&1 ); cat $O/k2_lane_address.txt | cut -c1-200
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
