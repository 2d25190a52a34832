#!/bin/bash
# round 6: the driver's command on one more box; usage: r06_boxes.sh <letter>
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=${1:-x}
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_cmd_box_$L.json 2> gpurun_out/r06_bench_driver_cmd_box_$L.err
python -c "
import json; d=json.load(open('gpurun_out/r06_bench_driver_cmd_box_$L.json')); l=d['legs']; c=d['config']; print('box $L: %.2f M rays/s, %.4f ms/step (window holds %d grid update(s); steady 200 steps %.4f), fox %.4f, hard %.4f, image %.4f, sdf %.4f ms/step, d2d %.0f GB/s, valu %.3f ns'%(d['value']/1e6,d['ms_per_step'],c['timed_window']['grid_updates_in_window'],c['steady_window']['ms_per_step'],l['fox']['ms_per_step'],l['hard']['ms_per_step'],l['image']['ms_per_step'],l['sdf']['ms_per_step'],c['calibration']['d2d_copy_GBps'],c['calibration']['valu_dependent_fma_ns']))"
