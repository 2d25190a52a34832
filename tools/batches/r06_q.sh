#!/bin/bash
# round 6, call q: what k_grad_bin's run merging is worth today (NGP_DEBUG_FLAGS 65536 = dense levels only, no merging on the hashed ones) and 2048-entry chunks (NGP_BIN_CHUNK_LOG2=11) once more,
# three passes; tests/test_sdf.py + test_pyngp.py (the knobs removed in this commit had SDF defaults; the snapshot test's new bar)
R=$PWD; O=gpurun_out/r06q; mkdir -p $O; . tools/batches/ab_lib.sh
timeout 600 python -m pytest tests/test_sdf.py tests/test_pyngp.py -q -x -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-300
for pass in 1 2 3; do
  ab_run default_p$pass NGP_X=1
  ab_run nohashedmerge_p$pass NGP_DEBUG_FLAGS=65536
  ab_run cl11_p$pass NGP_BIN_CHUNK_LOG2=11
done
