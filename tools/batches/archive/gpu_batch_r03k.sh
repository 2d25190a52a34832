#!/bin/bash
# round 3, call 12: error-proportional pixel sampling (error map, CDFs) -- kernel-level and trainer-level GPU tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03k
date
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item())"
timeout 900 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_train.py tests/test_gpu_dist.py -m gpu -q -p no:cacheprovider -k "error or rccl or k3 or prelaunched or tracks" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|FAILED|Error|assert" gpurun_out/${TAG}_pytest.log | cut -c1-400 | tail -25
date
timeout 300 python bench.py --no-cpu-baseline --no-fox-leg > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03k_bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['config'].get('calibration'))
print(d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])
PY
date
