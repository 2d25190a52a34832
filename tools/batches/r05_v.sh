#!/bin/bash
# round 5, call v (the round's last GPU seconds): distance items from the end of the batch first (the long walks must not start last) vs batch order; then tests/test_sdf.py with the new default
R=$PWD; O=gpurun_out/r05v; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in 1 0; do
  NGP_SDF_DIST_ORDER=$v timeout 30 python tools/f4_bench.py sdf > $O/f4_sdf_order_$v.jsonl 2> $O/f4_sdf_order_$v.err
  echo "dist order $v"; python -c "
import json
for l in open('$O/f4_sdf_order_$v.jsonl'):
    d=json.loads(l); print('   ', d['op'][:60], d['ms'])"
done
timeout 60 python -m pytest tests/test_sdf.py -q -x -m gpu > $O/pytest_sdf.log 2>&1; tail -2 $O/pytest_sdf.log | cut -c1-300
