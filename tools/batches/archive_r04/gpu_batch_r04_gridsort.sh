#!/bin/bash
# round 4: occupancy-grid update -- radix-sort key width of the density samples (begin bit) against the density inference's time; + the Testbed-level latent test
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_pyngp.py -m gpu -q -p no:cacheprovider -k "latents" > gpurun_out/r04_pytest_pyngp_latents.log 2>&1; tail -3 gpurun_out/r04_pytest_pyngp_latents.log
for rep in 1 2; do for bb in 6 10 13 16 99; do
  if [ $bb = 99 ]; then export NGP_DEBUG_FLAGS=4194304; else unset NGP_DEBUG_FLAGS; fi  # 99 = no sort at all (DBG_GRID_NO_SORT)
  NGP_GRID_SORT_BEGIN_BIT=$bb timeout 120 python bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0 --profile-steps 64 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_per_step']
print('begin_bit $bb rep $rep ms/step %.4f grid_misc %.4f density %.4f  (per update: %.0f + %.0f us)'%(d['ms_per_step'],k.get('occupancy_grid_misc',0),k.get('k_inference<density_only>',0),k.get('occupancy_grid_misc',0)*16e3,k.get('k_inference<density_only>',0)*16e3))"
done; done | tee gpurun_out/r04_grid_sort_bits.log
