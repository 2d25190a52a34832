#!/bin/bash
# round 6 (second session), call br: SDF step at 8 / 12 / 16 batches ahead, 100-step calls as well (the 20-step calls of f4_bench cut groups)
R=$PWD; O=gpurun_out/r06br; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for pass in 1 2; do for g in 8 12 16; do
  NGP_SDF_GROUP=$g timeout 60 python tools/f4_bench.py sdf > $O/sdf_g${g}_p$pass.jsonl 2> $O/sdf_g${g}_p$pass.err
  python -c "
import json
d=[json.loads(l) for l in open('$O/sdf_g${g}_p$pass.jsonl')]
print('group $g pass $pass: step', d[0]['ms'] if d else 'FAILED')"
done; done
