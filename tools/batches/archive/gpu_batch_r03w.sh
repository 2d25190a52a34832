#!/bin/bash
# round 3, call 24: K3 instance specialised for train_mode Nerf without depth supervision (107 registers instead of 128 + 64 B scratch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03w
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item())"
timeout 900 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_train.py -m gpu -q -p no:cacheprovider -k "k3 or converges or tracks or depth or rfl" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc $?"; tail -1 gpurun_out/${TAG}_pytest.log | cut -c1-300
run() { # label, env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --no-fox-leg --no-calibration > gpurun_out/${TAG}_bench_$label.json 2> gpurun_out/${TAG}_bench_$label.err
  python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_$label.json'))
k=d['roofline']['kernel_ms_per_step']
print('$label', round(d['ms_per_step'],4), round(d['value']/1e6,2), 'k3', k['k_compute_loss'], 'k2', k['k_inference'], 'scatter unit', k['k_train_fwd_bwd+k_grad_bin+k_grad_accumulate'])
PY
}
run plain NGP_X=1
run generic NGP_DEBUG_FLAGS_OR=1073741824
run plain2 NGP_X=1
run generic2 NGP_DEBUG_FLAGS_OR=1073741824
