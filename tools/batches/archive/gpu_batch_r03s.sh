#!/bin/bash
# round 3, call 20: T1 reads the encodings the lazy K2 left behind (K3 row -> sample map) instead of gathering them again
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03s
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item())"
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_dist.py tests/test_gpu_fox.py -m gpu -q -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/${TAG}_pytest.log | cut -c1-400 | tail -8
run() { # label, env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --no-fox-leg --no-calibration > gpurun_out/${TAG}_bench_$label.json 2> gpurun_out/${TAG}_bench_$label.err
  python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_$label.json'))
k=d['roofline']['kernel_ms_per_step']
print('$label', round(d['ms_per_step'],4), round(d['value']/1e6,2), 'k2', k['k_inference'], 'scatter unit', k['k_train_fwd_bwd+k_grad_bin+k_grad_accumulate'], 'k3', k['k_compute_loss'], 'frac', d['roofline']['frac'])
PY
}
run stash NGP_X=1
run gather NGP_DEBUG_FLAGS_OR=536870912
run stash2 NGP_X=1
run gather2 NGP_DEBUG_FLAGS_OR=536870912
