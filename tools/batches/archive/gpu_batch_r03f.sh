#!/bin/bash
# round 3, call 6: full GPU tier after depth supervision / rolling shutter / 16-bit Adam counters / colour-network depth; quick bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03f
date
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item(), torch.cuda.get_device_name(0))"
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|FAILED|mean rendered opacity|rolling shutter /|steady state" gpurun_out/${TAG}_pytest_gpu.log | cut -c1-330 | tail -24
date
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03f_bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['config'].get('calibration'), d.get('legs'))
print(d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])
PY
date
