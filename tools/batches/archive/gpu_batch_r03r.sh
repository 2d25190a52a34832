#!/bin/bash
# round 3, call 19: batched weight-fragment loads at block start (all MLP kernels); lazy K2 at 4 blocks per CU (128 registers, 28 B of scratch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03r
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item())"
NGP_K2_OCC4=1 timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py tests/test_encmlp.py -m gpu -q -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc $?"; tail -1 gpurun_out/${TAG}_pytest.log | cut -c1-300
run() { # label, env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --no-fox-leg --no-calibration > gpurun_out/${TAG}_bench_$label.json 2> gpurun_out/${TAG}_bench_$label.err
  python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_$label.json'))
k=d['roofline']['kernel_ms_per_step']
print('$label', round(d['ms_per_step'],4), 'k2', k['k_inference'], 'scatter unit', k['k_train_fwd_bwd+k_grad_bin+k_grad_accumulate'], 'w', k['k_wgrad'], 'grid inference', k['k_inference<density_only>'])
PY
}
run default NGP_X=1
run occ4 NGP_K2_OCC4=1
run default2 NGP_X=1
run occ4_2 NGP_K2_OCC4=1
run mult4 NGP_K2_GRID_MULT=4
run default3 NGP_X=1
