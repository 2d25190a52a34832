#!/bin/bash
# round 3, call 11: process-wide helper streams -- full GPU tier, default bench (fox leg in the same process), forced-DP bench, fp16 sum error
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03j
date
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item())"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/${TAG}_smoke.log | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|FAILED" gpurun_out/${TAG}_pytest.log | cut -c1-330 | tail -8
date
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03j_bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['config'].get('calibration'))
print(d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])
print('fox', d.get('legs',{}).get('fox')); print('cpu', d.get('cpu_baseline'))
PY
date
NGP_FORCE_DP=1 timeout 300 python bench.py --no-cpu-baseline --no-fox-leg > gpurun_out/${TAG}_bench_forced_dp.json 2> gpurun_out/${TAG}_bench_forced_dp.err; echo "forced dp rc $?"
python -c "
import json; d=json.load(open('gpurun_out/r03j_bench_forced_dp.json')); print('forced dp', d['value'], d['ms_per_step'])"
timeout 400 python tools/dp_fp16_sum_error.py 8 1000 > gpurun_out/${TAG}_dpsum.json 2> gpurun_out/${TAG}_dpsum.err; echo "dpsum rc $?"; cut -c1-900 gpurun_out/${TAG}_dpsum.json
date
