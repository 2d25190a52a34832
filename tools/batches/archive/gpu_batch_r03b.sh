#!/bin/bash
# round 3, call 2: what an RCCL communicator does to the process's kernels; the new network shapes; regression of the base.json kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03b
date
NGP_DP_FUSED_STEP=1 NCCL_DEBUG=INFO timeout 300 python tools/dp_diag2.py > gpurun_out/${TAG}_dp_diag2.json 2> gpurun_out/${TAG}_dp_diag2.err; echo "diag2 rc $?"
cat gpurun_out/${TAG}_dp_diag2.json | cut -c1-1200
grep -E "NCCL INFO" gpurun_out/${TAG}_dp_diag2.err | head -60 | cut -c1-220
date
timeout 900 python -m pytest tests/test_gpu_shapes.py -m gpu -q -x -s -p no:cacheprovider > gpurun_out/${TAG}_pytest_shapes.log 2>&1; echo "shapes rc $?"
grep -E "passed|failed|Error|error|assert" gpurun_out/${TAG}_pytest_shapes.log | tail -12 | cut -c1-400
date
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -m gpu -q -x -p no:cacheprovider > gpurun_out/${TAG}_pytest_model.log 2>&1; echo "model rc $?"
tail -5 gpurun_out/${TAG}_pytest_model.log | cut -c1-400
date
