#!/bin/bash
# round 3, call 16: k_grad_bin cursor atomic overlapped with the prefix sum and the LDS scatter (LDS-only barriers)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03o
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item())"
for T in 256 512; do NGP_BIN_THREADS=$T timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_shapes.py -m gpu -q -p no:cacheprovider > gpurun_out/${TAG}_pytest_$T.log 2>&1; echo "pytest $T rc $?"; tail -1 gpurun_out/${TAG}_pytest_$T.log | cut -c1-300; done
for T in 256 512 256 512; do
  NGP_BIN_THREADS=$T timeout 200 python bench.py --no-cpu-baseline --no-fox-leg --no-calibration > gpurun_out/${TAG}_bench_t$T.json 2> gpurun_out/${TAG}_bench_t$T.err
  python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_t$T.json'))
k=d['roofline']['kernel_ms_per_step']
print('bin threads $T', round(d['ms_per_step'],4), 'scatter unit', k['k_train_fwd_bwd+k_grad_bin+k_grad_accumulate'], 'k2', k['k_inference'], 'k1', k['k_generate_training_samples'], 'frac', d['roofline']['frac'])
PY
done
