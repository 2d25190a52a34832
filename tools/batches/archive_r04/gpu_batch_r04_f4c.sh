#!/bin/bash
# round 4: SDF ground truth, third version (while-while walks) and the number of stab rays walked by the per-point kernel
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_sdf.py -m gpu -q -p no:cacheprovider -s > gpurun_out/r04_pytest_sdf_c.log 2>&1; tail -3 gpurun_out/r04_pytest_sdf_c.log
for fr in 2 0 1 4; do echo "# NGP_SDF_FIRST_RAYS=$fr"; NGP_SDF_FIRST_RAYS=$fr timeout 100 python tools/f4_bench.py sdf 2>/dev/null; done > gpurun_out/r04_f4_bench_c.jsonl; cat gpurun_out/r04_f4_bench_c.jsonl
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r04_f4_prof_c -o f4 -- python tools/f4_bench.py sdf > /dev/null 2> gpurun_out/r04_f4_prof_c.err
