#!/bin/bash
# round 6, call e: lazy K2 with the level constants from an LDS table (NGP_K2_DEPTH 1: all gathers hoisted by the compiler; 2: two levels explicitly in flight) vs from GridMeta (0, rounds 1-5)
R=$PWD; O=gpurun_out/r06e; mkdir -p $O; . tools/batches/ab_lib.sh
for dp in 1 2; do NGP_K2_DEPTH=$dp timeout 300 python -m pytest tests/test_gpu_train.py -q -x -m gpu -k "lazy_k2 or t1_reuses or render_matches or training_loop" -p no:cacheprovider > $O/pytest_k2_$dp.log 2>&1; echo "depth $dp: $(tail -1 $O/pytest_k2_$dp.log | cut -c1-200)"; done
for pass in 1 2; do
  for dp in 0 1 2; do ab_run k2depth${dp}_p$pass NGP_K2_DEPTH=$dp; done
done
