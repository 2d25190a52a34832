#!/bin/bash
# round 4: occupancy-grid update with rocPRIM's Onesweep (MergeSortLimit = 0) instead of its merge sort; bit-exact grid tests; key width sweep
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "grid or occupancy" > gpurun_out/r04_pytest_grid_onesweep.log 2>&1; tail -3 gpurun_out/r04_pytest_grid_onesweep.log
for rep in 1 2; do for bb in 6 5 13 99; do
  if [ $bb = 99 ]; then export NGP_DEBUG_FLAGS=4194304; else unset NGP_DEBUG_FLAGS; fi  # 99 = no sort at all (DBG_GRID_NO_SORT)
  NGP_GRID_SORT_BEGIN_BIT=$bb timeout 120 python bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0 --profile-steps 64 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_per_step']
print('onesweep begin_bit $bb rep $rep ms/step %.4f grid_misc %.4f density %.4f  (per update: %.0f + %.0f us)'%(d['ms_per_step'],k.get('occupancy_grid_misc',0),k.get('k_inference<density_only>',0),k.get('occupancy_grid_misc',0)*16e3,k.get('k_inference<density_only>',0)*16e3))"
done; done | tee gpurun_out/r04_grid_sort_onesweep.log
unset NGP_DEBUG_FLAGS
for rep in 1 2 3; do timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('driver cmd rep $rep: %.4f ms/step %.2f M rays/s'%(d['ms_per_step'], d['value']/1e6))"; done | tee -a gpurun_out/r04_grid_sort_onesweep.log
