#!/bin/bash
# round 4: SDF ground truth, fourth version (distance query and first rays side by side: role by blockIdx.y)
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_sdf.py -m gpu -q -p no:cacheprovider -s > gpurun_out/r04_pytest_sdf_d.log 2>&1; tail -3 gpurun_out/r04_pytest_sdf_d.log
for fr in 3 2 4 6; do echo "# NGP_SDF_FIRST_RAYS=$fr"; NGP_SDF_FIRST_RAYS=$fr timeout 100 python tools/f4_bench.py sdf 2>/dev/null; done > gpurun_out/r04_f4_bench_d.jsonl; cat gpurun_out/r04_f4_bench_d.jsonl
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r04_f4_prof_d -o f4 -- python tools/f4_bench.py sdf > /dev/null 2> gpurun_out/r04_f4_prof_d.err
