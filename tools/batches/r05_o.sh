#!/bin/bash
# round 5, call o: the persistent SDF walker on 64-byte nodes (half boxes rounded outwards): parity (bit-identical to brute force), then interleaved against the
# 128-byte nodes (NGP_SDF_FULL_NODES=1) and round 4's three launches (NGP_SDF_PERSISTENT=0)
R=$PWD; O=gpurun_out/r05o; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_sdf.py -q -x -m gpu > $O/pytest_sdf.log 2>&1; tail -4 $O/pytest_sdf.log | cut -c1-600
for i in 1 2; do
  for v in half full r4; do
    case $v in half) E="NGP_X=1";; full) E="NGP_SDF_FULL_NODES=1";; r4) E="NGP_SDF_PERSISTENT=0";; esac
    env $E timeout 100 python tools/f4_bench.py sdf > $O/f4_sdf_${v}_$i.jsonl 2> $O/f4_sdf_${v}_$i.err
    echo "$v run $i"; python -c "
import json,sys
for l in open('$O/f4_sdf_${v}_$i.jsonl'):
    d=json.loads(l); print('   ', d['op'][:60], d['ms'])"
  done
done
