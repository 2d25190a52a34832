#!/bin/bash
# round 4, call i: level-per-XCD encoding of the round-0 tiles -- correctness tests + interleaved A/B
R=$PWD; O=gpurun_out/r04i; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -m pytest tests/test_gpu_train.py tests/test_gpu_nerf.py -q -s -k "lazy_k2 or t1_reuses or converges or tracks_oracle or fused_optimizer or k3_loss" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0"
for i in 1 2; do
  for v in xcd off byindex bpc2 bpc8; do
    case $v in xcd) E="NGP_X=1";; off) E="NGP_K2_XCD_ENCODE=0";; byindex) E="NGP_XCD_ENCODE_BY_BLOCK_INDEX=1";; bpc2) E="NGP_XCD_ENCODE_BLOCKS_PER_CU=2";; bpc8) E="NGP_XCD_ENCODE_BLOCKS_PER_CU=8";; esac
    env $E $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
k=d['roofline']['kernel_ms_per_step']
print("$v $i", round(d['ms_per_step'],4), round(d['value']/1e6,2), 'k2', k.get('k_inference'), 'enc', k.get('k_encode_tiles_xcd'), 'loss', round(d['config']['loss'],7))
PY
  done
done
