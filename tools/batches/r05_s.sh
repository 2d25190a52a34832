#!/bin/bash
# round 5, call s: the persistent SDF walker at 6 workgroups per CU (80 registers, 32 B scratch) against 4 (98 registers): parity, then interleaved A / B
R=$PWD; O=gpurun_out/r05s; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 200 python -m pytest tests/test_sdf.py -q -x -m gpu > $O/pytest_sdf.log 2>&1; tail -3 $O/pytest_sdf.log | cut -c1-400
for i in 1 2; do
  for v in 6 4; do
    NGP_SDF_WALK_OCC=$v timeout 100 python tools/f4_bench.py sdf > $O/f4_sdf_occ${v}_$i.jsonl 2> $O/f4_sdf_occ${v}_$i.err
    echo "occ $v run $i"; python -c "
import json
for l in open('$O/f4_sdf_occ${v}_$i.jsonl'):
    d=json.loads(l); print('   ', d['op'][:60], d['ms'])"
  done
done
