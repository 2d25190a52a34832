#!/bin/bash
# round 6 (second session), call bh: k1_count (multi-cascade marcher) as a two-phase kernel (a ray's chunk groups evaluated side by side by the workgroup's wavefronts; the exact-skip orbit of a
# chunk marked by pointer doubling instead of one scalar step per visited point): K1 parity tests, then the fox leg against the previous commit's library at 8 / 4 / 3 / 2 workgroups per CU
R=$PWD; O=gpurun_out/r06bh; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_fox.py tests/test_k1_lattice_model.py -q -x -m gpu -p no:cacheprovider -k "k1 or fox" > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-300
for pass in 1 2; do for v in prev new8 new4 new3 new2; do
  L="NGP_K1_MC_BLOCKS=${v#new}"; [ $v = prev ] && L="NGP_HIP_LIB=$R/gpurun_in/libngp_hip_prev.so"
  env $L timeout 300 python bench.py --gpus 1 --scene fox --pretrain 3000 --steps 200 --warmup 20 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --no-calibration --eval-views 0 --profile-steps 32 > $O/fox_${v}_p$pass.json 2> $O/fox_${v}_p$pass.err || tail -3 $O/fox_${v}_p$pass.err
  python - $O/fox_${v}_p$pass.json $v $pass <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "pass", sys.argv[3], round(d["ms_per_step"] * 1000, 1), "us/step", {k: round(v * 1000, 1) for k, v in d["roofline"].get("kernel_ms_per_step", {}).items()})
PY
done; done
