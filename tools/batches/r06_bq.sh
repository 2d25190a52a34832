#!/bin/bash
# round 6 (second session), call bq: three streams instead of four?  (NGP_NO_COMM_STREAM=1: the communication stream is not created; grid samples inside the update in both variants)
R=$PWD; O=gpurun_out/r06bq; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
. tools/batches/ab_lib.sh
for pass in 1 2 3; do
  ab_run four_p$pass NGP_DEBUG_FLAGS2_OR=4 NGP_NO_COMM_STREAM=0
  ab_run three_p$pass NGP_DEBUG_FLAGS2_OR=4 NGP_NO_COMM_STREAM=1
done
