#!/bin/bash
# round 6 (second session), call bi: what does the exact-skip part of k1_count cost on fox?  (DBG_K1_INDEPENDENT_LATTICE: every lattice point tested on its own -- no jump lengths, no walks)
R=$PWD; O=gpurun_out/r06bi; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for pass in 1 2; do for v in exact independent; do
  L="NGP_X=1"; [ $v = independent ] && L="NGP_DEBUG_FLAGS=$1"
  env $L timeout 300 python bench.py --gpus 1 --scene fox --pretrain 3000 --steps 200 --warmup 20 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --no-calibration --eval-views 0 --profile-steps 32 > $O/fox_${v}_p$pass.json 2> $O/fox_${v}_p$pass.err || tail -3 $O/fox_${v}_p$pass.err
  python - $O/fox_${v}_p$pass.json $v $pass <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "pass", sys.argv[3], round(d["ms_per_step"] * 1000, 1), "us/step", {k: round(v * 1000, 1) for k, v in d["roofline"].get("kernel_ms_per_step", {}).items()})
PY
done; done
