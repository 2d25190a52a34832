#!/bin/bash
# round 5, call t: the persistent SDF walker at 2 / 3 / 4 workgroups per CU (call s: 6 is slower than 4 -- is fewer faster still?)
R=$PWD; O=gpurun_out/r05t; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in 3 2 4; do
  NGP_SDF_WALK_OCC=$v timeout 100 python tools/f4_bench.py sdf > $O/f4_sdf_occ${v}.jsonl 2> $O/f4_sdf_occ${v}.err
  echo "occ $v"; python -c "
import json
for l in open('$O/f4_sdf_occ${v}.jsonl'):
    d=json.loads(l); print('   ', d['op'][:60], d['ms'])"
done
