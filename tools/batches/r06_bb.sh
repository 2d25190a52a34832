#!/bin/bash
# round 6 (second session), call bb: SDF trainer with the batches generated a GROUP per ground-truth launch, ahead and across calls (ngp_sdf_set_batches_ahead / NGP_SDF_GROUP):
# tests/test_sdf.py (bit-identity with the serial loop), then the step at group sizes 0 (serial) / 1 / 2 / 4 / 8 / 12 and 2 vs 4 walk workgroups per CU
R=$PWD; O=gpurun_out/r06bb; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests/test_sdf.py -q -x -s -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; grep -E "batches_ahead|passed|failed|Error|assert" $O/pytest.log | cut -c1-300 | tail -12
for pass in 1 2; do for v in "1 1 2" "0 1 2" "0 2 2" "0 4 2" "0 8 2" "0 12 2" "0 4 4" "0 8 4"; do set -- $v
  NGP_SDF_NO_PREFETCH=$1 NGP_SDF_GROUP=$2 NGP_SDF_WALK_OCC=$3 timeout 60 python tools/f4_bench.py sdf > $O/sdf_np$1_g$2_occ$3_p$pass.jsonl 2> $O/sdf_np$1_g$2_occ$3_p$pass.err
  python -c "
import json
d=[json.loads(l) for l in open('$O/sdf_np$1_g$2_occ$3_p$pass.jsonl')]
print('no_prefetch $1 group $2 occ $3 pass $pass: step', d[0]['ms'] if d else 'FAILED', 'ms; ground truth of one batch alone', d[1]['ms'] if len(d) > 1 else '-')"
done; done
