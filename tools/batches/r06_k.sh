#!/bin/bash
# round 6, call k: SDF step with the next batch's ground truth on a side stream under the training part (default) vs the serial loop (NGP_SDF_NO_PREFETCH=1), at 4 / 3 / 2 walk workgroups per CU;
# tests/test_sdf.py + tests/test_encmlp.py first
R=$PWD; O=gpurun_out/r06k; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_sdf.py tests/test_encmlp.py -q -x -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-300
for pass in 1 2; do for v in "1 4" "0 4" "0 3" "0 2"; do set -- $v
  NGP_SDF_NO_PREFETCH=$1 NGP_SDF_WALK_OCC=$2 timeout 60 python tools/f4_bench.py sdf > $O/sdf_np$1_occ$2_p$pass.jsonl 2> $O/sdf_np$1_occ$2_p$pass.err
  echo "no_prefetch $1 occ $2 pass $pass"; python -c "
import json
for l in open('$O/sdf_np$1_occ$2_p$pass.jsonl'):
    d=json.loads(l); print('   ', d['op'][:70], d['ms'])"
done; done
