#!/bin/bash
# Round-2 GPU batch N: tests + isolated kernel times (no overlap trace) + plain bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02n}
echo "== pytest" ; date
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|Error|error" gpurun_out/${TAG}_pytest_gpu.log | tail -12 | cut -c1-300
prof() { # tag, env...
  tag=$1; shift
  cd /tmp && rm -rf /tmp/prof_$tag && env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o t -- python $R/bench.py --pretrain 1000 --steps 100 --warmup 5 --no-cpu-baseline --eval-views 0 --profile-steps 0 > $R/gpurun_out/${TAG}_rocprof_$tag.log 2>&1; echo "rocprof $tag rc $?"
  cd $R
  find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats_$tag.csv \;
  T=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
  python tools/kernel_trace_summary.py "$T" > gpurun_out/${TAG}_kernel_trace_summary_$tag.txt 2>&1
  grep -v "at::native\|rocclr\|Cijk\|k_grid\|bitfield\|k_ema\|k_mark\|k_master\|k_build\|k_splat\|k_generate_grid" gpurun_out/${TAG}_kernel_trace_summary_$tag.txt | cut -c1-330
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/${TAG}_rocprof_$tag.log | head -2
}
echo "== traces"; date
prof nooverlap NGP_DEBUG_FLAGS=4096
echo "== bench" ; date
for i in 1 2; do
timeout 400 python bench.py --pretrain 1000 --steps 200 --warmup 5 --no-cpu-baseline --eval-views 0 --profile-steps 0 > gpurun_out/${TAG}_bench_$i.log 2>&1; echo "bench rc $?"
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_$i.log | head -2
done
date
