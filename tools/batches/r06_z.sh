#!/bin/bash
# round 6, call z: image step without its two memset launches (the optimizer sweep clears the grid gradients, the batch generator the loss word) and k_grad_accumulate at 256 threads for the
# image grid; tests/test_encmlp.py + test_sdf.py + test_ref_sdf (gpu part), then tools/f4_bench.py image against the previous commit's library
R=$PWD; O=gpurun_out/r06z; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_encmlp.py tests/test_sdf.py tests/test_pyngp.py -q -x -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-300
for pass in 1 2 3; do for v in prev new; do
  L=""; [ $v = prev ] && L="NGP_HIP_LIB=$R/gpurun_in/libngp_hip_prev.so"
  env $L timeout 100 python tools/f4_bench.py image > $O/f4_${v}_p$pass.jsonl 2> $O/f4_${v}_p$pass.err
  echo "$v pass $pass: $(python -c "
import json
print(' | '.join(str(json.loads(l)['ms']) for l in open('$O/f4_${v}_p$pass.jsonl')))")"
done; done
