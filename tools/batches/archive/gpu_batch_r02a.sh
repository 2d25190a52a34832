#!/bin/bash
# Round-2 GPU batch A: GPU parity tests, layout / K2 ablations, bench lines (synthetic lego-format + fox), rocprofv3 kernel stats.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest" ; date
timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r02a_pytest_gpu.log 2>&1; echo "pytest rc $?"
tail -15 gpurun_out/r02a_pytest_gpu.log
echo "== microbench" ; date
timeout 420 python tools/microbench.py 1000 32 > gpurun_out/r02a_microbench.log 2> gpurun_out/r02a_microbench.err; echo "microbench rc $?"
cat gpurun_out/r02a_microbench.log | cut -c1-600
echo "== bench" ; date
timeout 300 python bench.py --steps 100 --warmup 10 > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err; echo "bench rc $?"
cut -c1-1500 gpurun_out/r02a_bench.json
echo "== bench fox" ; date
timeout 420 python bench.py --scene fox --pretrain 3000 --steps 100 --warmup 10 --eval-views 3 --no-cpu-baseline > gpurun_out/r02a_bench_fox.json 2> gpurun_out/r02a_bench_fox.err; echo "fox rc $?"
cut -c1-1500 gpurun_out/r02a_bench_fox.json; tail -3 gpurun_out/r02a_bench_fox.err
echo "== rocprof" ; date
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o r02a -- python "$OLDPWD/bench.py" --pretrain 600 --steps 50 --warmup 5 --no-cpu-baseline --eval-views 0 > /tmp/prof_a.log 2>&1; echo "rocprof rc $?")
find /tmp/prof_a -name "*kernel_stats*" -exec cp {} gpurun_out/r02a_kernel_stats.csv \;
head -30 gpurun_out/r02a_kernel_stats.csv | cut -c1-200
date
