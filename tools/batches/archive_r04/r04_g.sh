#!/bin/bash
R=$PWD; O=gpurun_out/r04g; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -m pytest tests/test_gpu_nerf.py tests/test_gpu_model.py tests/test_gpu_train.py -q -s -k "k3_loss or training_step_gradients or tracks_oracle or error_proportional or lazy_k2 or converges" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0"
for i in 1 2 3; do
  for v in rpw4 rpw2; do
    case $v in rpw4) E="NGP_X=1";; rpw2) E="NGP_K3_RPW=2";; esac
    env $E $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
k=d['roofline']['kernel_ms_per_step']
print("$v $i", round(d['ms_per_step'],4), round(d['value']/1e6,2), 'k3', k.get('k_compute_loss'), 'psnr', d['config'].get('train_psnr_estimate_db'))
PY
  done
done
