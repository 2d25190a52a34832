#!/bin/bash
# Round-2 GPU batch W: selected tests + plain bench runs under env combinations (interleaved, twice) + isolated kernel times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02w}
TESTS=${2:-tests/test_gpu_model.py tests/test_gpu_train.py}
echo "== pytest $TESTS" ; date
timeout 900 python -m pytest $TESTS -m gpu -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|Error|error" gpurun_out/${TAG}_pytest_gpu.log | tail -6 | cut -c1-300
echo "== bench env combos"; date
for rep in 1 2; do
for cfg in "A=0" "NGP_T1_OCC=3" "NGP_BIN_SAMPLES=256" "NGP_T1_OCC=3 NGP_BIN_SAMPLES=256"; do
  env $cfg timeout 300 python bench.py --pretrain 1000 --steps 200 --warmup 5 --no-cpu-baseline --eval-views 0 --profile-steps 16 > gpurun_out/${TAG}_bench.tmp 2>&1
  python -c "
import json,sys
d=None
for line in open('gpurun_out/${TAG}_bench.tmp'):
    if line.startswith('{'): d=json.loads(line)
k=d['roofline']['kernel_ms_per_step']
print('%-40s ms/step %.4f rays/s %.2fM | T1+bin+acc %.4f K2 %.4f K1 %.4f' % ('$cfg', d['ms_per_step'], d['value']/1e6, k.get('k_train_fwd_bwd+k_grad_bin+k_grad_accumulate',0), k.get('k_inference',0), k.get('k_generate_training_samples',0)))"
done; done
prof() { # tag, env...
  tag=$1; shift
  cd /tmp && rm -rf /tmp/prof_$tag && env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o t -- python $R/bench.py --pretrain 1000 --steps 100 --warmup 5 --no-cpu-baseline --eval-views 0 --profile-steps 0 > $R/gpurun_out/${TAG}_rocprof_$tag.log 2>&1; echo "rocprof $tag rc $?"
  cd $R
  T=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
  python tools/kernel_trace_summary.py "$T" > gpurun_out/${TAG}_kernel_trace_summary_$tag.txt 2>&1
  grep -A16 "average step timeline" gpurun_out/${TAG}_kernel_trace_summary_$tag.txt | cut -c1-130
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/${TAG}_rocprof_$tag.log | head -2
  rm -rf /tmp/prof_$tag
}
echo "== trace"; date
prof nooverlap NGP_DEBUG_FLAGS=4096
date
