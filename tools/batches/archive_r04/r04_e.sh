#!/bin/bash
# round 4, call e: tightened parity tests + single-stream vs helper-stream step
R=$PWD; O=gpurun_out/r04e; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -m pytest tests/test_gpu_model.py tests/test_gpu_dist.py tests/test_gpu_train.py -q -s -k "training_step_gradients or tracks_oracle or fused_optimizer or rccl or multi_rank" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0"
for i in 1 2; do
  for v in streams single; do
    case $v in streams) E="NGP_X=1";; single) E="NGP_DEBUG_FLAGS_OR=4096";; esac
    env $E $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
print("$v $i", round(d['ms_per_step'],4), round(d['value']/1e6,2))
PY
  done
done
