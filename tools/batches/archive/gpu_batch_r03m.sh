#!/bin/bash
# round 3, call 14: lazy K2 tile width 8 vs 16 (fewer evaluated samples behind the cut vs more sequential tiles per ray)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03m
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item())"
for T in 16 8 16 8; do
  NGP_K2_TILE=$T timeout 200 python bench.py --no-cpu-baseline --no-fox-leg --no-calibration > gpurun_out/${TAG}_bench_t$T.json 2> gpurun_out/${TAG}_bench_t$T.err
  python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_t$T.json'))
k=d['roofline']['kernel_ms_per_step']
print('tile $T', round(d['ms_per_step'],4), 'k2', k['k_inference'], 'k1', k['k_generate_training_samples'], 't1', k['k_train_fwd_bwd+k_grad_bin+k_grad_accumulate'], 'k3', k['k_compute_loss'], d['config'].get('network_evaluations_per_step'))
PY
done
