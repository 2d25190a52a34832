#!/bin/bash
# round 6 (second session), call bm: the occupancy-grid update's samples drawn + sorted ahead on a side stream (default) vs inside the update (NGP_DEBUG_FLAGS2_OR=4):
# the bit-identity test + the grid / training tests, then headline (20-step driver window and 400 steps) and fox, interleaved
R=$PWD; O=gpurun_out/r06bm; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_dist.py -q -x -s -m gpu -p no:cacheprovider -k "grid_samples_drawn_ahead or world1" > $O/pytest_ahead.log 2>&1; grep -E "grid samples ahead|passed|failed|Error|assert" $O/pytest_ahead.log | cut -c1-300 | tail -6
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_nerf.py -q -x -m gpu -p no:cacheprovider > $O/pytest_train.log 2>&1; tail -2 $O/pytest_train.log | cut -c1-300
. tools/batches/ab_lib.sh
for pass in 1 2 3; do
  ab_run inside_p$pass NGP_DEBUG_FLAGS2_OR=4
  ab_run ahead_p$pass NGP_DEBUG_FLAGS2_OR=0
done
for pass in 1 2; do for v in 1 0; do
  NGP_DEBUG_FLAGS2_OR=$((v * 4)) timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --no-calibration --eval-views 0 --steady-steps 200 > $O/drv_noahead${v}_p$pass.json 2> $O/drv_noahead${v}_p$pass.err
  NGP_DEBUG_FLAGS2_OR=$((v * 4)) timeout 300 python bench.py --gpus 1 --scene fox --pretrain 3000 --steps 200 --warmup 20 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --no-calibration --eval-views 0 > $O/fox_noahead${v}_p$pass.json 2> $O/fox_noahead${v}_p$pass.err
  python - $O/drv_noahead${v}_p$pass.json $O/fox_noahead${v}_p$pass.json $v $pass <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); f = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("no_ahead", sys.argv[3], "pass", sys.argv[4], "driver window", round(d["ms_per_step"] * 1000, 1), "us/step, steady 200 steps", round(d["config"]["steady_window"]["ms_per_step"] * 1000, 1), "| fox", round(f["ms_per_step"] * 1000, 1))
PY
done; done
