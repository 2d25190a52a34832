#!/bin/bash
# round 6, call a: state at the start of the round -- smoke, the driver's bench command, kernel trace (no overlap) of 200 steps
R=$PWD; O=gpurun_out/r06a; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-3000 $O/bench.json
cd /tmp && rm -rf /tmp/prof_a && NGP_DEBUG_FLAGS=4096 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o t -- python $R/bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0 > $R/$O/rocprof.log 2>&1; echo "rocprof rc $?"
cd $R
T=$(find /tmp/prof_a -name "*kernel_trace.csv" | head -1)
python tools/kernel_trace_summary.py "$T" > $O/kernel_trace_summary_nooverlap.txt 2>&1
grep -A18 "average step timeline" $O/kernel_trace_summary_nooverlap.txt | cut -c1-130
