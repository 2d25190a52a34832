#!/bin/bash
# round 5: the driver's command on one more box; usage: r05_boxes.sh <letter>
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=${1:-x}
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_cmd_box_$L.json 2> gpurun_out/r05_bench_driver_cmd_box_$L.err
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_driver_cmd_box_$L.json')); l=d['legs']; print('box $L: %.2f M rays/s, %.4f ms/step, fox %.4f, image %.4f, sdf %.4f ms/step, valu %.3f ns'%(d['value']/1e6,d['ms_per_step'],l['fox']['ms_per_step'],l['image']['ms_per_step'],l['sdf']['ms_per_step'],d['config']['calibration']['valu_dependent_fma_ns']))"
