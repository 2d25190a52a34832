#!/bin/bash
# round 3, call 25: k_grad_bin with 1024 samples / threads per block (longer runs, half the cursor atomics, 4 wavefronts per SIMD)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03x
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item())"
NGP_BIN_THREADS=1024 timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc $?"; tail -1 gpurun_out/${TAG}_pytest.log | cut -c1-300
run() { # label, env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --no-fox-leg --no-calibration > gpurun_out/${TAG}_bench_$label.json 2> gpurun_out/${TAG}_bench_$label.err
  python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_$label.json'))
k=d['roofline']['kernel_ms_per_step']
print('$label', round(d['ms_per_step'],4), round(d['value']/1e6,2), 'scatter unit', k['k_train_fwd_bwd+k_grad_bin+k_grad_accumulate'], 'frac', d['roofline']['frac'])
PY
}
run t512 NGP_X=1
run t1024 NGP_BIN_THREADS=1024
run t512_2 NGP_X=1
run t1024_2 NGP_BIN_THREADS=1024
