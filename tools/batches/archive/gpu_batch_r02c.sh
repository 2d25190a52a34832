#!/bin/bash
# Round-2 GPU batch C: tests (dist fixes, run.py drop-in), kernel traces with timeline (default / no stream overlap / K2 16-wide tiles),
# MFMA + traffic counters at the trained steady state
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest" ; date
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r02c_pytest_gpu.log 2>&1; echo "pytest rc $?"
grep -v "^$" gpurun_out/r02c_pytest_gpu.log | tail -30 | cut -c1-300
prof() { # tag, env...
  tag=$1; shift
  cd /tmp && rm -rf /tmp/prof_$tag && env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o t -- python $R/bench.py --pretrain 1000 --steps 100 --warmup 5 --no-cpu-baseline --eval-views 0 --profile-steps 0 > $R/gpurun_out/r02c_rocprof_$tag.log 2>&1; echo "rocprof $tag rc $?"
  cd $R
  find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02c_kernel_stats_$tag.csv \;
  T=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
  python tools/kernel_trace_summary.py "$T" > gpurun_out/r02c_kernel_trace_summary_$tag.txt 2>&1
  cut -c1-330 gpurun_out/r02c_kernel_trace_summary_$tag.txt
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r02c_rocprof_$tag.log | head -2
}
echo "== rocprof traces" ; date
prof default NGP_X=1
prof nooverlap NGP_DEBUG_FLAGS=4096
prof k2tile16 NGP_K2_TILE=16 NGP_K2_ROUNDS=4 NGP_DEBUG_FLAGS=4096
echo "== pmc mfma + traffic" ; date
timeout 600 python tools/pmc_probe.py $R/gpurun_out/r02c_pmc 1000 8 default mfma,tcc2 > gpurun_out/r02c_pmc_probe.log 2>&1; tail -40 gpurun_out/r02c_pmc_probe.log | cut -c1-250
timeout 900 bash tools/pmc_traffic.sh $R/gpurun_out/r02c_pmc_traffic > gpurun_out/r02c_pmc_traffic.log 2>&1; tail -45 gpurun_out/r02c_pmc_traffic.log | cut -c1-200
date
