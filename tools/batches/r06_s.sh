#!/bin/bash
# round 6, call s: lazy K2 with x-adjacent corner pairs as one 16-byte gather per lane where the table makes them neighbours (NGP_K2_DEPTH=3: two levels in flight, 4: one; both spill at 168 registers)
R=$PWD; O=gpurun_out/r06s; mkdir -p $O; . tools/batches/ab_lib.sh
NGP_K2_DEPTH=4 timeout 400 python -m pytest tests/test_gpu_train.py -q -x -m gpu -k "lazy_k2 or t1_reuses or training_loop" -p no:cacheprovider > $O/pytest_4.log 2>&1; echo "pairs: $(tail -1 $O/pytest_4.log | cut -c1-200)"
for pass in 1 2; do
  ab_run prod_p$pass NGP_X=1
  ab_run pairs3_p$pass NGP_K2_DEPTH=3
  ab_run pairs4_p$pass NGP_K2_DEPTH=4
done
