#!/bin/bash
# round 3, call 13: lazy K2 with more blocks than resident slots (dynamic hand-out of tiles by the dispatcher)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03l
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item())"
for M in 1 2 4 8 16 1 4; do
  NGP_K2_GRID_MULT=$M timeout 200 python bench.py --no-cpu-baseline --no-fox-leg --no-calibration > gpurun_out/${TAG}_bench_m$M.json 2> gpurun_out/${TAG}_bench_m$M.err
  python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_m$M.json'))
k=d['roofline']['kernel_ms_per_step']
print('mult $M', round(d['ms_per_step'],4), 'k2', k['k_inference'], 'k1', k['k_generate_training_samples'], 't1', k['k_train_fwd_bwd+k_grad_bin+k_grad_accumulate'])
PY
done
