#!/bin/bash
# round 6 (second session), call bu: the SDF step measured in ONE continuous host-clock window of 320 steps (event pairs around 20-step calls left part of the side stream's work outside
# the window): serial loop and 1 / 4 / 8 / 12 / 16 batches ahead, two passes
R=$PWD; O=gpurun_out/r06bu; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for pass in 1 2; do for v in "1 1" "0 1" "0 4" "0 8" "0 12" "0 16"; do set -- $v
  NGP_SDF_NO_PREFETCH=$1 NGP_SDF_GROUP=$2 timeout 90 python tools/f4_bench.py sdf > $O/sdf_np$1_g$2_p$pass.jsonl 2> $O/sdf_np$1_g$2_p$pass.err
  python -c "
import json
d=[json.loads(l) for l in open('$O/sdf_np$1_g$2_p$pass.jsonl')]
print('no_prefetch $1 group $2 pass $pass: step', d[0]['ms'] if d else 'FAILED', 'ms (320 steps, one window); ground truth of one batch alone', d[1]['ms'] if len(d) > 1 else '-')"
done; done
