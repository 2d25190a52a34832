#!/bin/bash
# round 6 (second session), call bc: k_grad_accumulate requests its first U x THREADS records together with the cursor (one dependent round trip less per block) and reads the
# table's old gradients only when its list overflowed.  Parity tests, then the image / SDF legs (tools/f4_bench.py) and the headline step against the previous commit's library.
R=$PWD; O=gpurun_out/r06bc; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_encmlp.py tests/test_sdf.py tests/test_gpu_model.py tests/test_gpu_shapes.py -q -x -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_train.py -q -x -m gpu -p no:cacheprovider -k "fused or optimizer or tracks" > $O/pytest_train.log 2>&1; tail -2 $O/pytest_train.log | cut -c1-300
for pass in 1 2; do for v in prev new; do
  L="NGP_X=1"; [ $v = prev ] && L="NGP_HIP_LIB=$R/gpurun_in/libngp_hip_prev.so"
  env $L timeout 100 python tools/f4_bench.py > $O/f4_${v}_p$pass.jsonl 2> $O/f4_${v}_p$pass.err
  echo "$v pass $pass"; python -c "
import json
print('   ', ' | '.join(str(json.loads(l)['ms']) for l in open('$O/f4_${v}_p$pass.jsonl')), '  (sdf step | sdf ground truth, batch half | ground truth 2^18 uniform | image step 2^16 | image step 2^18)')"
done; done
. tools/batches/ab_lib.sh
for pass in 1 2 3; do
  ab_run prev_p$pass NGP_HIP_LIB=$R/gpurun_in/libngp_hip_prev.so
  ab_run new_p$pass NGP_X=1
done
