#!/bin/bash
# round 6, call aa: the one red test of the final batch's tier (a wrong assertion in this round's own tests/test_bench_scenes.py) again, then the driver's command on this box
R=$PWD; O=gpurun_out/r06aa; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_bench_scenes.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-300
bash tools/batches/r06_boxes.sh a
