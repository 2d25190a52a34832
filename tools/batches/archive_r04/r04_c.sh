#!/bin/bash
# round 4, call c: fused optimizer epilogue -- equivalence test, interleaved A/B with the driver's command, kernel trace
R=$PWD; O=gpurun_out/r04c; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -m pytest tests/test_gpu_train.py -q -s -k "fused_optimizer or growing_dataset or render_matches" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0"
for i in 1 2 3; do
  for v in fused unfused fused11 ; do
    case $v in fused) E="NGP_X=1";; unfused) E="NGP_NO_FUSED_ADAM=1";; fused11) E="NGP_BIN_CHUNK_LOG2=11";; esac
    env $E $B > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${v}_$i.json") if l.startswith('{')][-1])
k=d['roofline']['kernel_ms_per_step']
print("$v $i", round(d['ms_per_step'],4), round(d['value']/1e6,2), {a:k[a] for a in list(k)[:6]})
PY
  done
done
cd /tmp && rm -rf /tmp/prof_c && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o t -- python $R/bench.py --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0 > $R/$O/rocprof.log 2>&1; cd $R
T=$(find /tmp/prof_c -name "*kernel_trace.csv" | head -1)
python tools/kernel_trace_summary.py "$T" > $O/kernel_trace_summary_overlap_fused.txt 2>&1
grep -A16 "average step timeline" $O/kernel_trace_summary_overlap_fused.txt | cut -c1-130
grep "steady-state" $O/kernel_trace_summary_overlap_fused.txt | cut -c1-400
