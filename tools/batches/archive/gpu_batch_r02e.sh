#!/bin/bash
# Round-2 GPU batch E: full GPU test suite (encmlp / image trainers, multi-rank, run.py drop-in)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02e}
timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?"
grep -v "^$" gpurun_out/${TAG}_pytest_gpu.log | grep -v "Warning\|warn\|amdgpu.ids\|Gloo\|socket.cpp" | tail -60 | cut -c1-400
