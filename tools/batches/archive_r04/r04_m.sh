#!/bin/bash
# round 4, call m: k1_count_segments / k1_write_list (segment prepass + sample lists) vs the chunk kernels
R=$PWD; O=gpurun_out/r04m; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_nerf.py tests/test_gpu_train.py -q -x -k "k1 or converges or tracks_oracle or dist or sample_cap" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python tools/microbench.py 600 32 default,k1_chunk_march,default_again > $O/microbench.log 2>&1; cat $O/microbench.log | cut -c1-900
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0"
for i in 1 2 3; do
  for v in seg chunk seg6; do
    case $v in seg) E="NGP_X=1";; chunk) E="NGP_DEBUG_FLAGS=33554432";; seg6) E="NGP_K1_SEG_BLOCKS_PER_CU=3";; esac
    env $E timeout 200 $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
k=d['roofline']['kernel_ms_per_step']
print("$v $i", round(d['ms_per_step'],4), round(d['value']/1e6,2), {n:v for n,v in k.items() if 'k1' in n or 'generate' in n}, 'marched', d['config']['marched_samples_last_step'], 'loss', round(d['config']['loss'],7))
PY
  done
done
