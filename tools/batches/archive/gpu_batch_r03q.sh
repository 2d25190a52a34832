#!/bin/bash
# round 3, call 18: bin / accumulate shape sweep with the 512-thread k_grad_bin (chunk 2^11 vs 2^12)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03q
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item())"
run() { # label, env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --no-fox-leg --no-calibration > gpurun_out/${TAG}_bench_$label.json 2> gpurun_out/${TAG}_bench_$label.err
  python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_$label.json'))
k=d['roofline']['kernel_ms_per_step']
print('$label', round(d['ms_per_step'],4), 'scatter unit', k['k_train_fwd_bwd+k_grad_bin+k_grad_accumulate'], 'frac', d['roofline']['frac'])
PY
}
run default NGP_X=1
run chunk11 NGP_BIN_CHUNK_LOG2=11
run default2 NGP_X=1
run chunk11_256thr NGP_BIN_CHUNK_LOG2=11 NGP_BIN_THREADS=256
run default3 NGP_X=1
