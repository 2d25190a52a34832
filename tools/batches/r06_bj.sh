#!/bin/bash
# round 6 (second session), call bj: (1) k1_count's replay asks the NEXT group's first record at a group's last chunk (a quarter of all chunks were walked only because the
# one-wavefront kernel did not know the next chunk yet): K1 parity tests + fox A/B against the previous commit; (2) the snapshot-continuation test (six loss samples, bar on the mean) five times
R=$PWD; O=gpurun_out/r06bj; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_fox.py tests/test_k1_lattice_model.py -q -x -m gpu -p no:cacheprovider -k "k1 or fox" > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-300
for i in 1 2 3 4 5; do timeout 300 python -m pytest tests/test_pyngp.py -q -x -s -m gpu -p no:cacheprovider -k "snapshot_written_elsewhere" > $O/pytest_snap_$i.log 2>&1; grep -E "continued 96|passed|failed" $O/pytest_snap_$i.log | cut -c1-260; done
for pass in 1 2 3; do for v in prev new; do
  L="NGP_X=1"; [ $v = prev ] && L="NGP_HIP_LIB=$R/gpurun_in/libngp_hip_prev.so"
  env $L timeout 300 python bench.py --gpus 1 --scene fox --pretrain 3000 --steps 200 --warmup 20 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --no-calibration --eval-views 0 --profile-steps 32 > $O/fox_${v}_p$pass.json 2> $O/fox_${v}_p$pass.err || tail -3 $O/fox_${v}_p$pass.err
  python - $O/fox_${v}_p$pass.json $v $pass <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "pass", sys.argv[3], round(d["ms_per_step"] * 1000, 1), "us/step", {k: round(v * 1000, 1) for k, v in d["roofline"].get("kernel_ms_per_step", {}).items()})
PY
done; done
