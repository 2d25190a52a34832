#!/bin/bash
# round 6 (second session), call be: rocprofv3 kernel trace of the fox leg (aabb_scale 4: the multi-cascade K1 kernels k1_count<8, false> / k1_write)
R=$PWD; O=$R/gpurun_out/r06be; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp && rm -rf /tmp/prof_fox && NGP_K1_MC_BLOCKS=${MC:-4} NGP_DEBUG_FLAGS=4096 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fox -o t -- python $R/bench.py --gpus 1 --scene fox --pretrain 3000 --steps 200 --warmup 20 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --no-calibration --eval-views 0 --steady-steps 0 > $O/rocprof_fox.log 2>&1; echo "rocprof rc $?"
cd $R
find /tmp/prof_fox -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_fox.csv \;
T=$(find /tmp/prof_fox -name "*kernel_trace.csv" | head -1)
python tools/kernel_trace_summary.py "$T" > $O/kernel_trace_summary_fox.txt 2>&1
head -24 $O/kernel_trace_summary_fox.txt | cut -c1-150
grep -A16 "average step timeline" $O/kernel_trace_summary_fox.txt | cut -c1-130
