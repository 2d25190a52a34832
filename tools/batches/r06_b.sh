#!/bin/bash
# round 6, call b: where the next step's K1 forks off the backward pass (NGP_K1_FORK 0 = behind K4 (today), 1 = behind k_train_fused, 2 = behind k_grad_bin = beside k_grad_accumulate),
# with 4096- and 2048-entry chunks (128 / 64 KiB accumulate blocks); interleaved, two passes
R=$PWD; O=gpurun_out/r06b; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
B="python bench.py --gpus 1 --steps 400 --warmup 20 --no-cpu-baseline --no-fox-leg --no-f4-legs --no-calibration --eval-views 0 --profile-steps 0"
for pass in 1 2; do
for v in "0 12" "1 12" "2 12" "0 11" "2 11"; do
  set -- $v
  NGP_K1_FORK=$1 NGP_BIN_CHUNK_LOG2=$2 timeout 120 $B > $O/fork$1_cl$2_p$pass.json 2> $O/fork$1_cl$2_p$pass.err
  echo "fork $1 chunk_log2 $2 pass $pass: $(grep -o '"ms_per_step": [0-9.]*' $O/fork$1_cl$2_p$pass.json | head -1)"
done; done
