#!/bin/bash
# round 5, call m: code size as a lever -- k1_setup<PLAIN> (37 -> 17 KiB of instructions) and k_compute_loss_v2<2, false, PLAIN, TGT> (32 -> 12 KiB, 107 -> 96 registers):
# parity (two trainers, general instances forced on one of them: bit-identical parameters; fox = OpenCV lens), then the driver's command interleaved with the general
# instances (NGP_DEBUG_FLAGS2_OR=2 NGP_DEBUG_FLAGS_OR=2^30)
R=$PWD; O=gpurun_out/r05m; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 500 python -m pytest tests/test_gpu_train.py -q -x -m gpu -k "small_instances or training_loop_tracks" > $O/pytest_train.log 2>&1; tail -5 $O/pytest_train.log | cut -c1-900
timeout 300 python -m pytest tests/test_gpu_fox.py -q -x -m gpu > $O/pytest_fox.log 2>&1; tail -3 $O/pytest_fox.log | cut -c1-600
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-f4-legs --no-calibration --eval-views 0"
for i in 1 2 3; do
  for v in new general; do
    case $v in new) E="NGP_X=1";; general) E="NGP_DEBUG_FLAGS2_OR=2 NGP_DEBUG_FLAGS_OR=1073741824";; esac
    env $E timeout 150 $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
k=d['roofline']['kernel_ms_per_step']
print("$v $i", round(d['ms_per_step'],4), round(d['value']/1e6,2), {a:round(b*1000,1) for a,b in k.items()}, 'loss', round(d['config']['loss'],7))
PY
  done
done
