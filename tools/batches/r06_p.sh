#!/bin/bash
# round 6, call p: r06_o again with the dynamic tile hand-out of the lazy K2 repaired (workgroups past n_tiles / 8 returned before taking their range: the first training steps lost tiles)
R=$PWD; O=gpurun_out/r06p; mkdir -p $O; . tools/batches/ab_lib.sh
timeout 900 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_train.py tests/test_gpu_model.py -q -x -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-300
for pass in 1 2; do
  ab_run prev_p$pass NGP_HIP_LIB=$R/gpurun_in/libngp_hip_prev.so
  ab_run new_p$pass NGP_X=1
  ab_run k3occ5_p$pass NGP_K3_OCC=5
  ab_run k1writelds_p$pass NGP_K1_WRITE_LDS=1
  ab_run k2static_p$pass NGP_K2_STATIC=1
done
