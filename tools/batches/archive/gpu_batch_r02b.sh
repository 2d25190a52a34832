#!/bin/bash
# Round-2 GPU batch B: tests incl. the multi-rank ones, K1 fix check, kernel trace with per-round K2 timing, fox bench, small A/B PSNR
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest" ; date
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r02b_pytest_gpu.log 2>&1; echo "pytest rc $?"
tail -25 gpurun_out/r02b_pytest_gpu.log | cut -c1-400
echo "== microbench" ; date
timeout 420 python tools/microbench.py 1000 32 default,k1_independent_lattice,k2_tile32_r3,k2_tile16_r3,default_again > gpurun_out/r02b_microbench.log 2> gpurun_out/r02b_microbench.err; echo "microbench rc $?"
cut -c1-700 gpurun_out/r02b_microbench.log
echo "== rocprof" ; date
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o r02b -- python $R/bench.py --pretrain 1000 --steps 100 --warmup 5 --no-cpu-baseline --eval-views 0 --profile-steps 0 > $R/gpurun_out/r02b_rocprof.log 2>&1; echo "rocprof rc $?"
cd $R
find /tmp/prof_b -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02b_kernel_stats.csv \;
T=$(find /tmp/prof_b -name "*kernel_trace.csv" | head -1)
python tools/kernel_trace_summary.py "$T" > gpurun_out/r02b_kernel_trace_summary.txt 2>&1; cat gpurun_out/r02b_kernel_trace_summary.txt | cut -c1-200
tail -3 gpurun_out/r02b_rocprof.log | cut -c1-300
echo "== bench fox" ; date
timeout 500 python bench.py --scene fox --pretrain 3000 --steps 100 --warmup 10 --eval-views 3 --no-cpu-baseline > gpurun_out/r02b_bench_fox.json 2> gpurun_out/r02b_bench_fox.err; echo "fox rc $?"
cut -c1-2500 gpurun_out/r02b_bench_fox.json; tail -3 gpurun_out/r02b_bench_fox.err
echo "== A/B psnr (small)" ; date
timeout 600 python bench.py --pretrain 200 --steps 20 --warmup 5 --no-cpu-baseline --eval-views 4 --eval-res 400 --ab-psnr 1000,3000 --profile-steps 4 > gpurun_out/r02b_bench_ab_small.json 2> gpurun_out/r02b_bench_ab_small.err; echo "ab rc $?"
python -c "import json;d=json.load(open('gpurun_out/r02b_bench_ab_small.json'));print(json.dumps(d['config'].get('ab_psnr')))"
date
