#!/bin/bash
# A/B PSNR spread at 5000 steps: production vs reference order, repeated; and with the dense levels' half atomics (round-2a path)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
TAG=${1:-r02ab}
for i in 1 2 3; do
  for cfg in "A=0" "NGP_DEBUG_FLAGS_OR=8388608"; do
    env $cfg timeout 600 python bench.py --pretrain 200 --steps 20 --warmup 5 --no-cpu-baseline --eval-views 8 --eval-res 800 --eval-spp 8 --ab-psnr 5000 --profile-steps 4 > gpurun_out/${TAG}_ab_$i.json 2> gpurun_out/${TAG}_ab_$i.err
    python -c "import json;d=json.load(open('gpurun_out/${TAG}_ab_$i.json'));a=d['config']['ab_psnr'];print('$cfg', a['production'], a['reference_order'], a['delta_db'])"
  done
done
date
