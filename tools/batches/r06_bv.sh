#!/bin/bash
# round 6 (second session), call bv: k1_count keeps only the first max_mip + 2 (>= 4) levels of the coarse occupancy in LDS (fox: 16 KiB instead of 32 -> four workgroups per CU
# instead of three; higher levels read the coarse bit from memory): K1 parity tests, fox leg against the previous commit's library
R=$PWD; O=gpurun_out/r06bv; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_fox.py tests/test_k1_lattice_model.py -q -x -m gpu -p no:cacheprovider -k "k1 or fox" > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-300
for pass in 1 2 3; do for v in prev new; do
  L="NGP_X=1"; [ $v = prev ] && L="NGP_HIP_LIB=$R/gpurun_in/libngp_hip_prev.so"
  env $L timeout 300 python bench.py --gpus 1 --scene fox --pretrain 3000 --steps 200 --warmup 20 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --no-calibration --eval-views 0 --profile-steps 32 > $O/fox_${v}_p$pass.json 2> $O/fox_${v}_p$pass.err || tail -3 $O/fox_${v}_p$pass.err
  python - $O/fox_${v}_p$pass.json $v $pass <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "pass", sys.argv[3], round(d["ms_per_step"] * 1000, 1), "us/step", {k: round(v * 1000, 1) for k, v in list(d["roofline"].get("kernel_ms_per_step", {}).items())[:3]})
PY
done; done
