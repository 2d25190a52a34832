#!/bin/bash
# round 6, call v: k_grad_bin with one sample per thread (512 threads) also for 2048-entry chunks -- the default since r06_q ran the two-samples-per-thread instance there
R=$PWD; O=gpurun_out/r06v; mkdir -p $O; . tools/batches/ab_lib.sh
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-300
for pass in 1 2 3; do
  ab_run bin256_p$pass NGP_BIN_THREADS_256=1
  ab_run bin512_p$pass NGP_X=1
done
