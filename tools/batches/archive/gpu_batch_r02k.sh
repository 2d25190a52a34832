#!/bin/bash
# Round-2 GPU batch K: tests + microbench variants + plain bench runs (overlap on), no profiler
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02k}
VARS=${2:-default,k2_rounds3,t1_dense_inline,default_again}
echo "== pytest" ; date
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|Error|error" gpurun_out/${TAG}_pytest_gpu.log | tail -12 | cut -c1-300
echo "== microbench" ; date
timeout 420 python tools/microbench.py 1000 32 $VARS > gpurun_out/${TAG}_microbench.log 2> gpurun_out/${TAG}_microbench.err; echo "microbench rc $?"
cut -c1-900 gpurun_out/${TAG}_microbench.log
echo "== bench" ; date
for f in 0 262144; do
NGP_DEBUG_FLAGS=$f timeout 400 python bench.py --pretrain 1000 --steps 200 --warmup 5 --no-cpu-baseline --eval-views 0 --profile-steps 0 > gpurun_out/${TAG}_bench_flags$f.log 2>&1; echo "bench flags=$f rc $?"
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_flags$f.log | head -2
done
NGP_K2_ROUNDS=3 timeout 400 python bench.py --pretrain 1000 --steps 200 --warmup 5 --no-cpu-baseline --eval-views 0 --profile-steps 0 > gpurun_out/${TAG}_bench_k2r3.log 2>&1; echo "bench k2 rounds 3 rc $?"
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_k2r3.log | head -2
date
