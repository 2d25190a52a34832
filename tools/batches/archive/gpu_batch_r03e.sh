#!/bin/bash
# round 3, call 5: depth supervision tests, the re-toleranced lens / dist tests, fox seed study for the notebook pin
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03e
date
timeout 1500 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_train.py tests/test_gpu_lens.py tests/test_gpu_dist.py tests/test_gpu_fox.py -m gpu -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|FAILED|rendered depth|rccl world-1|steady state" gpurun_out/${TAG}_pytest.log | cut -c1-330 | tail -20
date
timeout 900 python tools/fox_notebook_pin.py 2000 gpurun_out/${TAG}_fox_notebook_pin.json 8 > gpurun_out/${TAG}_fox_pin.log 2>&1; echo "fox pin rc $?"
python -c "
import json;d=json.load(open('gpurun_out/${TAG}_fox_notebook_pin.json'))
for k in ('l16f2','l8f4_current_base_json'):
    print(k,[round(x['loss_tail_mean'],6) for x in d['runs'][k]], d[k+'_summary'])"
date
