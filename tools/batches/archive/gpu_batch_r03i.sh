#!/bin/bash
# round 3, call 10: k1_count coarse bracket -- exactness tests + ablation; fp16 sum error with the right per-rank counts
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03i
date
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item())"
timeout 1200 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_fox.py tests/test_gpu_train.py tests/test_gpu_ab_psnr.py -m gpu -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|FAILED|held-out" gpurun_out/${TAG}_pytest.log | cut -c1-330 | tail -6
date
timeout 600 python tools/microbench.py 1000 32 default,k1_no_coarse_range,k1_round2,default_again > gpurun_out/${TAG}_microbench.log 2> gpurun_out/${TAG}_microbench.err; echo "microbench rc $?"
cut -c1-330 gpurun_out/${TAG}_microbench.log
date
timeout 300 python bench.py --no-cpu-baseline --no-fox-leg > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03i_bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['config'].get('calibration'))
print(d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])
PY
date
