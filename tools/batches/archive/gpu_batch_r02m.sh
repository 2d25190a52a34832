#!/bin/bash
# Round-2 GPU batch M: statistics of test_training_loop_tracks_oracle under ablations + K1 experiments (isolated kernel times)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02m}
echo "== tracking test x4 per config"; date
for cfg in "A=0" "NGP_DEBUG_FLAGS_OR=524288" "NGP_K2_ROUNDS=3" "NGP_DEBUG_FLAGS_OR=524288 NGP_K2_ROUNDS=3"; do
  for i in 1 2 3; do
    env $cfg timeout 300 python -m pytest tests/test_gpu_train.py -m gpu -q -s -p no:cacheprovider -k test_training_loop_tracks_oracle 2>&1 | grep -E "^4 [0-9]+ |passed|failed" | tr '\n' ' '; echo " [$cfg]"
  done
done
prof() { # tag, env...
  tag=$1; shift
  cd /tmp && rm -rf /tmp/prof_$tag && env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o t -- python $R/bench.py --pretrain 1000 --steps 100 --warmup 5 --no-cpu-baseline --eval-views 0 --profile-steps 0 > $R/gpurun_out/${TAG}_rocprof_$tag.log 2>&1; echo "rocprof $tag rc $?"
  cd $R
  T=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
  python tools/kernel_trace_summary.py "$T" > gpurun_out/${TAG}_kernel_trace_summary_$tag.txt 2>&1
  grep "k1_\|k_compute_loss\|k_fill\|optimizer steps" gpurun_out/${TAG}_kernel_trace_summary_$tag.txt | cut -c1-330
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/${TAG}_rocprof_$tag.log | head -2
}
echo "== K1 experiments (no overlap: isolated kernel times)"; date
prof base NGP_DEBUG_FLAGS=4096
prof nopixel NGP_DEBUG_FLAGS=4096 NGP_K1_EXP=1
prof setup_early_exit NGP_DEBUG_FLAGS=4096 NGP_K1_EXP=3
prof group16 NGP_DEBUG_FLAGS=4096 NGP_K1_EXP=8
date
