#!/bin/bash
# round 6 (second session), call bk: the GPU tier three times in a row on one box (flake hunt: the driver runs it once with -x)
R=$PWD; O=gpurun_out/r06bk; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for i in 1 2 3; do
  timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu_$i.log 2>&1; echo "run $i rc $?"; grep -E "^FAILED|passed|failed" $O/pytest_gpu_$i.log | tail -5 | cut -c1-300
done
