#!/bin/bash
# round 6, call g: tcnn's two EMA kernels (half-precision state = the inference buffer by default; "full_precision" = fp32 state fed with the masters) instead of the hybrid of rounds 1-5;
# fused epilogue without the half-parameter read.  Tests that touch the optimizer / snapshots / data parallel, then the A/B against the previous commit's library (built beside this one).
R=$PWD; O=gpurun_out/r06g; mkdir -p $O; . tools/batches/ab_lib.sh
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py tests/test_pyngp.py tests/test_encmlp.py tests/test_gpu_dist.py tests/test_run_py_dropin.py -q -x -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "$(tail -3 $O/pytest.log | cut -c1-300)"
for pass in 1 2 3; do
  ab_run old_p$pass NGP_HIP_LIB=$R/gpurun_in/libngp_hip_prev.so
  ab_run new_p$pass NGP_X=1
done
