#!/bin/bash
# round 6, call c: k1_count_segments with 8 wavefronts per workgroup (NGP_K1_SEG_WAVES=8, new default) vs 4; K1 bit-identity tests first
R=$PWD; O=gpurun_out/r06c; mkdir -p $O; . tools/batches/ab_lib.sh
timeout 300 python -m pytest tests/test_gpu_nerf.py -q -x -m gpu -k "k1" -p no:cacheprovider > $O/pytest_k1.log 2>&1; tail -3 $O/pytest_k1.log | cut -c1-300
for pass in 1 2; do
  ab_run w4_p$pass NGP_K1_SEG_WAVES=4
  ab_run w8_p$pass NGP_K1_SEG_WAVES=8
  ab_run w8_cl11_p$pass NGP_K1_SEG_WAVES=8 NGP_BIN_CHUNK_LOG2=11
done
