#!/bin/bash
# Round-2 GPU batch Q: one test file + microbench variants + plain bench with env overrides
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02q}
VARS=${2:-default,k2_tile16,k2_tile32,default_again}
echo "== pytest (train)" ; date
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|Error|error" gpurun_out/${TAG}_pytest_gpu.log | tail -12 | cut -c1-300
echo "== microbench" ; date
timeout 420 python tools/microbench.py 1000 32 $VARS > gpurun_out/${TAG}_microbench.log 2> gpurun_out/${TAG}_microbench.err; echo "microbench rc $?"
cut -c1-900 gpurun_out/${TAG}_microbench.log
echo "== bench" ; date
for t in 32 16 32 16; do
NGP_K2_TILE=$t timeout 400 python bench.py --pretrain 1000 --steps 200 --warmup 5 --no-cpu-baseline --eval-views 0 --profile-steps 0 > gpurun_out/${TAG}_bench_tile$t.log 2>&1; echo "bench tile=$t rc $?"
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_tile$t.log | head -2
done
date
