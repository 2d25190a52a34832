#!/bin/bash
# round 4: SDF ground truth, sixth version (query points along a Morton curve) against batch order on the same box
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_sdf.py -m gpu -q -p no:cacheprovider -s > gpurun_out/r04_pytest_sdf_g.log 2>&1; tail -3 gpurun_out/r04_pytest_sdf_g.log
for v in sorted unsorted sorted unsorted; do echo "# $v"; if [ $v = unsorted ]; then export NGP_SDF_NO_POINT_SORT=1; else unset NGP_SDF_NO_POINT_SORT; fi; timeout 100 python tools/f4_bench.py sdf 2>/dev/null; done > gpurun_out/r04_f4_bench_g.jsonl; cut -c1-70,200-420 gpurun_out/r04_f4_bench_g.jsonl
unset NGP_SDF_NO_POINT_SORT
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r04_f4_prof_g -o f4 -- python tools/f4_bench.py sdf > /dev/null 2> gpurun_out/r04_f4_prof_g.err
