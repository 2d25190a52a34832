#!/bin/bash
# round 6, call l: image / SDF model kernels with the level constants from LDS and without the per-corner 32-bit modulo; k_grad_accumulate zeroes only the accumulators a chunk owns and leaves at once
# when its list is empty; k_encmlp_wgrad_reduce with 16 wavefronts.  Parity tests, then tools/f4_bench.py (image + sdf) against the previous commit's library.
R=$PWD; O=gpurun_out/r06l; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_encmlp.py tests/test_sdf.py tests/test_gpu_model.py tests/test_gpu_shapes.py -q -x -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-300
for pass in 1 2; do for v in prev new; do
  L=""; [ $v = prev ] && L="NGP_HIP_LIB=$R/gpurun_in/libngp_hip_prev.so"
  env $L timeout 100 python tools/f4_bench.py > $O/f4_${v}_p$pass.jsonl 2> $O/f4_${v}_p$pass.err
  echo "$v pass $pass"; python -c "
import json
for l in open('$O/f4_${v}_p$pass.jsonl'):
    d=json.loads(l); print('   ', d['op'][:90], d['ms'])"
done; done
