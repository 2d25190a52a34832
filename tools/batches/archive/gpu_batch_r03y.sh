#!/bin/bash
# round 3, call 30: k1_count / k1_write persistent grid vs resident blocks (k1_count: 77 registers = 6 blocks of 4 wavefronts per CU, the grid asked for 8)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03y
python -c "import torch; x=torch.ones(1<<24,device='cuda'); print('gpu sanity', x.sum().item())"
run() { # label, env...
  label=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --no-fox-leg --no-calibration > gpurun_out/${TAG}_bench_$label.json 2> gpurun_out/${TAG}_bench_$label.err
  python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_$label.json'))
k=d['roofline']['kernel_ms_per_step']
print('$label', round(d['ms_per_step'],4), round(d['value']/1e6,2), 'k1', k['k_generate_training_samples'], 'k3', k['k_compute_loss'])
PY
}
run b8 NGP_X=1
run b6 NGP_K1_BLOCKS_PER_CU=6
run b5 NGP_K1_BLOCKS_PER_CU=5
run b4 NGP_K1_BLOCKS_PER_CU=4
run b12 NGP_K1_BLOCKS_PER_CU=12
run b8_2 NGP_X=1
run b6_2 NGP_K1_BLOCKS_PER_CU=6
