#!/bin/bash
# round 3, call 8: k1_count first-point skip -- exactness tests + ablation
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03h
date
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item())"
timeout 1200 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_fox.py tests/test_gpu_train.py -m gpu -q -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc $?"
tail -4 gpurun_out/${TAG}_pytest.log | cut -c1-300
date
timeout 600 python tools/microbench.py 1000 32 default,k1_no_first_point_skip,default_again,k1_no_first_point_skip > gpurun_out/${TAG}_microbench.log 2> gpurun_out/${TAG}_microbench.err; echo "microbench rc $?"
cut -c1-420 gpurun_out/${TAG}_microbench.log
date
