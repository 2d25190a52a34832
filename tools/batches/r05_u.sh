#!/bin/bash
# round 5, call u: how often a walking stab ray polls its point's mark (device-scope load per lane): every 2nd round (mask 1, so far), every 8th (7), never (0xffffffff);
# then tests/test_sdf.py with the new default (7)
R=$PWD; O=gpurun_out/r05u; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in 7 1 0xffffffff; do
  NGP_SDF_POLL_MASK=$v timeout 60 python tools/f4_bench.py sdf > $O/f4_sdf_poll_$v.jsonl 2> $O/f4_sdf_poll_$v.err
  echo "poll mask $v"; python -c "
import json
for l in open('$O/f4_sdf_poll_$v.jsonl'):
    d=json.loads(l); print('   ', d['op'][:60], d['ms'])"
done
timeout 100 python -m pytest tests/test_sdf.py -q -x -m gpu > $O/pytest_sdf.log 2>&1; tail -2 $O/pytest_sdf.log | cut -c1-300
