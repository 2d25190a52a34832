#!/bin/bash
# round 6, call f: level constants from the LDS table in the eager k_inference too (density-grid update, renderer) + lazy K2 default depth 2; model / train / shapes parity tests, then A/B against NGP_K2_DEPTH=0
R=$PWD; O=gpurun_out/r06f; mkdir -p $O; . tools/batches/ab_lib.sh
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_shapes.py -q -x -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "$(tail -1 $O/pytest.log | cut -c1-200)"
for pass in 1 2; do
  ab_run depth0_p$pass NGP_K2_DEPTH=0
  ab_run depth2_p$pass NGP_K2_DEPTH=2
done
