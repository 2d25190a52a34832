#!/bin/bash
# round 4: SDF ground truth, second version (two-box nodes, batched leaves)
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_sdf.py -m gpu -q -p no:cacheprovider -s > gpurun_out/r04_pytest_sdf_b.log 2>&1; tail -3 gpurun_out/r04_pytest_sdf_b.log
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r04_f4_prof_b -o f4 -- python tools/f4_bench.py sdf > gpurun_out/r04_f4_bench_b.jsonl 2> gpurun_out/r04_f4_prof_b.err; cat gpurun_out/r04_f4_bench_b.jsonl
