#!/bin/bash
# round 6 (second session), call bo: k1_count_segments' LDS staging array out of scratch (96 B per lane written + read back by every thread: an array of HIP uint4 structs was never
# promoted to registers; found with hipcc -Rpass-analysis=kernel-resource-usage): K1 parity tests, then the headline against the previous commit's library
R=$PWD; O=gpurun_out/r06bo; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_nerf.py tests/test_k1_lattice_model.py tests/test_gpu_train.py -q -x -m gpu -p no:cacheprovider -k "k1 or tracks or lattice" > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-300
. tools/batches/ab_lib.sh
for pass in 1 2 3; do
  ab_run prev_p$pass NGP_HIP_LIB=$R/gpurun_in/libngp_hip_prev.so
  ab_run new_p$pass NGP_X=1
done
