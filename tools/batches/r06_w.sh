#!/bin/bash
# round 6, call w: k_grad_bin's run merging through DPP row shifts (runs cut at the 16-lane rows) instead of ds_bpermute shuffles; parity (every scatter layout vs the oracle), then A/B
R=$PWD; O=gpurun_out/r06w; mkdir -p $O; . tools/batches/ab_lib.sh
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_shapes.py tests/test_gpu_train.py -q -x -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-300
for pass in 1 2 3; do
  ab_run prev_p$pass NGP_HIP_LIB=$R/gpurun_in/libngp_hip_prev.so
  ab_run dpp_p$pass NGP_X=1
done
