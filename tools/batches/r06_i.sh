#!/bin/bash
# round 6, call i: kernel trace (no stream overlap) of 200 steps at the current head
R=$PWD; O=gpurun_out/r06i; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp && rm -rf /tmp/prof_a && NGP_DEBUG_FLAGS=4096 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o t -- python $R/bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-fox-leg --no-f4-legs --no-calibration --eval-views 0 > $R/$O/rocprof.log 2>&1; echo "rocprof rc $?"
cd $R
T=$(find /tmp/prof_a -name "*kernel_trace.csv" | head -1)
python tools/kernel_trace_summary.py "$T" > $O/kernel_trace_summary_nooverlap.txt 2>&1
grep -A18 "average step timeline" $O/kernel_trace_summary_nooverlap.txt | cut -c1-130
grep -B2 -A30 "occupancy-grid update" $O/kernel_trace_summary_nooverlap.txt | cut -c1-150 | head -60
