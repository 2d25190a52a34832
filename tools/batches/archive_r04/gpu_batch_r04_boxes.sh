#!/bin/bash
# round 4: the driver's command on one more box (+ the data-parallel test after its bound was restated); usage: gpu_batch_r04_boxes.sh <letter> [dist]
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=${1:-x}
if [ "${2:-}" = dist ]; then timeout 400 python -m pytest tests/test_gpu_dist.py -m gpu -q -p no:cacheprovider -s > gpurun_out/r04_pytest_gpu_dist.log 2>&1; grep -E "steady state|passed|failed" gpurun_out/r04_pytest_gpu_dist.log | cut -c1-250; fi
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_cmd_box_$L.json 2> gpurun_out/r04_bench_driver_cmd_box_$L.err
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_driver_cmd_box_$L.json')); print('box $L: %.2f M rays/s, %.4f ms/step, fox leg %.4f ms/step, valu %.3f ns'%(d['value']/1e6,d['ms_per_step'],d['legs']['fox']['ms_per_step'],d['config']['calibration']['valu_dependent_fma_ns']))"
