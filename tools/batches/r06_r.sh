#!/bin/bash
# round 6, call r: lazy K2 with slot refill (NGP_K2_REFILL=1: a finished ray's slot takes the next ray's first tile at once) at 16- and 8-sample tiles vs the production kernel
R=$PWD; O=gpurun_out/r06r; mkdir -p $O; . tools/batches/ab_lib.sh
for tw in 16 8; do NGP_K2_REFILL=1 NGP_K2_TILE=$tw timeout 400 python -m pytest tests/test_gpu_train.py -q -x -m gpu -k "lazy_k2 or t1_reuses or render_matches or training_loop or fused_optimizer" -p no:cacheprovider > $O/pytest_$tw.log 2>&1; echo "refill tile $tw: $(tail -1 $O/pytest_$tw.log | cut -c1-200)"; done
for pass in 1 2 3; do
  ab_run prod_p$pass NGP_X=1
  ab_run refill16_p$pass NGP_K2_REFILL=1 NGP_K2_TILE=16
  ab_run refill8_p$pass NGP_K2_REFILL=1 NGP_K2_TILE=8
done
python - <<'PY'
import json
for v in ("prod","refill16","refill8"):
    d=json.loads(open(f"gpurun_out/r06r/{v}_p1.json").read().strip().splitlines()[-1]); c=d["config"]
    print(v, "network evaluations", c["network_evaluations_last_step"], "marched", c["marched_samples_last_step"], "rays hit", c["rays_hit_last_step"], "loss", c["loss"])
PY
