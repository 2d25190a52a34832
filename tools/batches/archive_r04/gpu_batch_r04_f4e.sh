#!/bin/bash
# round 4: the image / SDF trainers' encoding backward through the record lists
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_encmlp.py tests/test_sdf.py -m gpu -q -p no:cacheprovider -s > gpurun_out/r04_pytest_encmlp_lists.log 2>&1; tail -5 gpurun_out/r04_pytest_encmlp_lists.log
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r04_f4_prof_e -o f4 -- python tools/f4_bench.py > gpurun_out/r04_f4_bench_e.jsonl 2> gpurun_out/r04_f4_prof_e.err; cut -c1-100,150-330 gpurun_out/r04_f4_bench_e.jsonl
