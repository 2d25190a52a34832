#!/bin/bash
# round 4: SDF ground truth (two-kernel BVH walk) -- tests, rates, per-kernel table; the extra-dims trainer tests
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_sdf.py tests/test_extra_dims.py -m gpu -q -p no:cacheprovider -s > gpurun_out/r04_pytest_sdf_extra.log 2>&1; tail -3 gpurun_out/r04_pytest_sdf_extra.log
timeout 200 python tools/f4_bench.py > gpurun_out/r04_f4_bench.jsonl 2> gpurun_out/r04_f4_bench.err; cat gpurun_out/r04_f4_bench.jsonl; tail -3 gpurun_out/r04_f4_bench.err
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r04_f4_prof -o f4 -- python tools/f4_bench.py sdf > /dev/null 2> gpurun_out/r04_f4_prof.err
ls gpurun_out/r04_f4_prof | head; find gpurun_out/r04_f4_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} > gpurun_out/r04_f4_kernel_stats_sdf.csv; cat gpurun_out/r04_f4_kernel_stats_sdf.csv'
