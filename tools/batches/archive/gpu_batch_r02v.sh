#!/bin/bash
# Round-2 GPU batch V: selected test files + microbench variants
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02v}
TESTS=${2:-tests/test_gpu_model.py}
VARS=${3:-default,bin_dense_levels,default_again}
echo "== pytest $TESTS" ; date
timeout 900 python -m pytest $TESTS -m gpu -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|Error|error|bin_dense" gpurun_out/${TAG}_pytest_gpu.log | tail -12 | cut -c1-600
echo "== microbench" ; date
timeout 420 python tools/microbench.py 1000 32 $VARS > gpurun_out/${TAG}_microbench.log 2> gpurun_out/${TAG}_microbench.err; echo "microbench rc $?"
cut -c1-900 gpurun_out/${TAG}_microbench.log
date
