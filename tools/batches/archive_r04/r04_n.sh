#!/bin/bash
# round 4, call n: isolated kernel times of the new K1 (no stream overlap), workgroups per CU
R=$PWD; O=gpurun_out/r04n; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_nerf.py -q -x -k "k1 or sample_cap or k3_loss" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for v in 4 chunk; do
  case $v in chunk) E="NGP_DEBUG_FLAGS=4096|33554432";; *) E="NGP_DEBUG_FLAGS=4096 NGP_K1_SEG_BLOCKS_PER_CU=$v";; esac
  E=$(echo $E | sed 's/4096|33554432/33558528/')
  cd /tmp && rm -rf /tmp/prof_n && env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_n -o t -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fox-leg --no-calibration --eval-views 0 > $R/$O/rocprof_$v.log 2>&1; cd $R
  T=$(find /tmp/prof_n -name "*kernel_trace.csv" | head -1)
  python tools/kernel_trace_summary.py "$T" > $O/kernel_trace_summary_nooverlap_$v.txt 2>&1
  echo "== $v"; grep "^k1\|^k_inference_tiles\|^k_grad\|^k_wgrad2\|^k_compute\|^k_train" $O/kernel_trace_summary_nooverlap_$v.txt | cut -c1-120
done
