#!/bin/bash
# round 3, call 21: stash T1 without spills (scheduling barrier behind the stash loads), 3 vs 4 blocks per CU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03t
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item())"
for O in 3 4; do NGP_T1_STASH_OCC=$O timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -p no:cacheprovider -k "reuses or converges or tracks" > gpurun_out/${TAG}_pytest_$O.log 2>&1; echo "pytest occ $O rc $?"; tail -1 gpurun_out/${TAG}_pytest_$O.log | cut -c1-300; done
run() { # label, env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --no-fox-leg --no-calibration > gpurun_out/${TAG}_bench_$label.json 2> gpurun_out/${TAG}_bench_$label.err
  python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_$label.json'))
k=d['roofline']['kernel_ms_per_step']
print('$label', round(d['ms_per_step'],4), round(d['value']/1e6,2), 'k2', k['k_inference'], 'scatter unit', k['k_train_fwd_bwd+k_grad_bin+k_grad_accumulate'], 'frac', d['roofline']['frac'])
PY
}
run gather NGP_DEBUG_FLAGS_OR=536870912
run stash3 NGP_X=1
run stash4 NGP_T1_STASH_OCC=4
run gather2 NGP_DEBUG_FLAGS_OR=536870912
run stash3_2 NGP_X=1
run stash4_2 NGP_T1_STASH_OCC=4
