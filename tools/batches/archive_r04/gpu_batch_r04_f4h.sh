#!/bin/bash
# round 4: SDF ground truth, query points by (cost class, Morton curve) vs batch order vs plain Morton order, on one box
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_sdf.py tests/test_pyngp.py -m gpu -q -p no:cacheprovider -k "sdf" > gpurun_out/r04_pytest_sdf_h0.log 2>&1; tail -1 gpurun_out/r04_pytest_sdf_h0.log
NGP_SDF_POINT_ORDER=1 timeout 200 python -m pytest tests/test_sdf.py -m gpu -q -p no:cacheprovider -s > gpurun_out/r04_pytest_sdf_h1.log 2>&1; tail -1 gpurun_out/r04_pytest_sdf_h1.log
for v in 0 1 2 0 1; do echo "# NGP_SDF_POINT_ORDER=$v"; NGP_SDF_POINT_ORDER=$v timeout 100 python tools/f4_bench.py sdf 2>/dev/null; done > gpurun_out/r04_f4_bench_h.jsonl
python - <<'P'
import json
for l in open('gpurun_out/r04_f4_bench_h.jsonl'):
    if l.startswith('#'): print(l.strip()); continue
    d=json.loads(l); print('   ', d['op'][:46], 'ms', d['ms'])
P
