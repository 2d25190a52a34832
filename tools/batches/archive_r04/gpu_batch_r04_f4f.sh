#!/bin/bash
# round 4: SDF ground truth, fifth version (4-wide nodes of one 128-byte line)
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_sdf.py -m gpu -q -p no:cacheprovider -s > gpurun_out/r04_pytest_sdf_f.log 2>&1; tail -3 gpurun_out/r04_pytest_sdf_f.log
for fr in 4 2 6; do echo "# NGP_SDF_FIRST_RAYS=$fr"; NGP_SDF_FIRST_RAYS=$fr timeout 100 python tools/f4_bench.py sdf 2>/dev/null; done > gpurun_out/r04_f4_bench_f.jsonl; cut -c1-70,200-420 gpurun_out/r04_f4_bench_f.jsonl
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r04_f4_prof_f -o f4 -- python tools/f4_bench.py sdf > /dev/null 2> gpurun_out/r04_f4_prof_f.err
