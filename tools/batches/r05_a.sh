#!/bin/bash
# round 5, call a: the new parity tests (extra dims through K1 / K3 / K4 at stride 7 + n, the one-rank RCCL identity, the latents' rng order / rebuilds), then the driver's command with the image + sdf legs
R=$PWD; O=gpurun_out/r05a; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_extra_dims_kernels.py -q -x -m gpu > $O/pytest_extra_kernels.log 2>&1; tail -15 $O/pytest_extra_kernels.log | cut -c1-600
timeout 600 python -m pytest tests/test_extra_dims.py -q -s -m gpu -k "training_step_gradients" > $O/pytest_extra_grad.log 2>&1; grep -E "^(3|16) |reference-order|passed|failed|Error|assert" $O/pytest_extra_grad.log | cut -c1-700 | tail -20
timeout 600 python -m pytest tests/test_gpu_dist.py -q -x -s -m gpu -k "world1" > $O/pytest_dist_w1.log 2>&1; tail -12 $O/pytest_dist_w1.log | cut -c1-600
timeout 900 python -m pytest tests/test_pyngp.py tests/test_extra_dims.py -q -x -m gpu -k "latent or extra or snapshot_carries" > $O/pytest_latents.log 2>&1; tail -12 $O/pytest_latents.log | cut -c1-600
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -3 $O/bench_driver_cmd.err | cut -c1-300
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_driver_cmd.json") if l.startswith('{')][-1])
print(round(d['ms_per_step'],4), round(d['value']/1e6,2), json.dumps(d['roofline']['kernel_ms_per_step']))
print(json.dumps(d.get('legs'))[:3000])
PY
