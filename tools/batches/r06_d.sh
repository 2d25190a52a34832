#!/bin/bash
# round 6, call d: K3 workgroup shape (NGP_K3_SHAPE: 0 = 16 wavefronts per workgroup, grid 512 (today); 161 = the same with grid 256 = resident; 85 / 86 = 8 wavefronts at 5 / 6 per SIMD;
# 45 / 46 = 4 wavefronts at 5 / 6 per SIMD), interleaved; K3 parity tests under the candidate shapes first
R=$PWD; O=gpurun_out/r06d; mkdir -p $O; . tools/batches/ab_lib.sh
for sh in 85 46; do NGP_K3_SHAPE=$sh timeout 300 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_train.py -q -x -m gpu -k "k3 or render_matches or training_loop" -p no:cacheprovider > $O/pytest_k3_$sh.log 2>&1; echo "shape $sh: $(tail -1 $O/pytest_k3_$sh.log | cut -c1-200)"; done
for pass in 1 2; do
  for sh in 0 161 85 86 45 46; do ab_run k3shape${sh}_p$pass NGP_K3_SHAPE=$sh; done
done
