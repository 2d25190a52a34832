#!/bin/bash
# round 3, first call: the data-parallel test with the all-reduced loss, the headline line on this box, and the world-size-1 overhead diagnosis
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03a
echo "== nproc $(nproc)"; date
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest_dist.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|steady state|loss after" gpurun_out/${TAG}_pytest_dist.log | tail -8
date
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc $?"; cut -c1-900 gpurun_out/${TAG}_bench.json
date
: > gpurun_out/${TAG}_dp_diag.jsonl
run() { name=$1; comm=$2; shift 2; env "$@" timeout 300 python tools/dp_diag.py $name $comm 2>gpurun_out/${TAG}_dp_diag_$name.err | tail -1 | tee -a gpurun_out/${TAG}_dp_diag.jsonl | cut -c1-600; }
run single nocomm NGP_X=1
run split nocomm NGP_TRAIN_SPLIT_PHASES=1
run comm_fused comm NGP_DP_FUSED_STEP=1
run comm_split comm NGP_X=1
run comm_split_skip comm NGP_DP_SKIP_ALLREDUCE=1
run single_again nocomm NGP_X=1
date
