#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02dpt}
cd /tmp && rm -rf /tmp/prof_dp && NGP_FORCE_DP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dp -o t -- python $R/bench.py --pretrain 1000 --steps 100 --warmup 5 --no-cpu-baseline --eval-views 0 --profile-steps 0 --dp-backend rccl > $R/gpurun_out/${TAG}_rocprof.log 2>&1; echo "rc $?"
cd $R
T=$(find /tmp/prof_dp -name "*kernel_trace.csv" | head -1)
python tools/kernel_trace_summary.py "$T" > gpurun_out/${TAG}_kernel_trace_summary.txt 2>&1
grep -v "at::native\|Cijk" gpurun_out/${TAG}_kernel_trace_summary.txt | head -30 | cut -c1-130
grep -A20 "average step timeline" gpurun_out/${TAG}_kernel_trace_summary.txt | cut -c1-130
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/${TAG}_rocprof.log | head -2
