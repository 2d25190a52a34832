#!/bin/bash
# round 4: smoke + the whole GPU tier at the final HEAD
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1 | tee gpurun_out/r04_smoke_final.log
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_pytest_gpu_final.log 2>&1; tail -3 gpurun_out/r04_pytest_gpu_final.log
