#!/bin/bash
# Round-2 GPU batch D: tests, K2 bulk-append check (microbench + per-round trace)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02d}
echo "== pytest" ; date
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?"
grep -v "^$" gpurun_out/${TAG}_pytest_gpu.log | tail -12 | cut -c1-300
echo "== microbench" ; date
timeout 420 python tools/microbench.py 1000 32 default,w_single_role,bin_no_merge,separate_grad_memset,round1_backward,default_again > gpurun_out/${TAG}_microbench.log 2> gpurun_out/${TAG}_microbench.err; echo "microbench rc $?"
cut -c1-700 gpurun_out/${TAG}_microbench.log
prof() { # tag, env...
  tag=$1; shift
  cd /tmp && rm -rf /tmp/prof_$tag && env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o t -- python $R/bench.py --pretrain 1000 --steps 100 --warmup 5 --no-cpu-baseline --eval-views 0 --profile-steps 0 > $R/gpurun_out/${TAG}_rocprof_$tag.log 2>&1; echo "rocprof $tag rc $?"
  cd $R
  find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats_$tag.csv \;
  T=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
  python tools/kernel_trace_summary.py "$T" > gpurun_out/${TAG}_kernel_trace_summary_$tag.txt 2>&1
  grep -v "at::native\|rocclr\|Cijk" gpurun_out/${TAG}_kernel_trace_summary_$tag.txt | head -24 | cut -c1-110; tail -2 gpurun_out/${TAG}_kernel_trace_summary_$tag.txt | cut -c1-400
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/${TAG}_rocprof_$tag.log | head -2
}
echo "== rocprof traces" ; date
prof nooverlap NGP_DEBUG_FLAGS=4096
prof overlap NGP_X=1
date
