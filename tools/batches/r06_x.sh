#!/bin/bash
# round 6, call x: does a K1 with a small LDS footprint hide better under the backward pass?  chunk kernels (NGP_DEBUG_FLAGS 33554432: k1_count<8, true> / k1_write, 4 KiB of LDS, slower alone)
# vs the segment kernels (36 KiB of LDS per workgroup), helper streams on (the timed step) and off (4096 added: no overlap)
R=$PWD; O=gpurun_out/r06x; mkdir -p $O; . tools/batches/ab_lib.sh
for pass in 1 2; do
  ab_run seg_overlap_p$pass NGP_X=1
  ab_run chunk_overlap_p$pass NGP_DEBUG_FLAGS=33554432
  ab_run seg_nooverlap_p$pass NGP_DEBUG_FLAGS=4096
  ab_run chunk_nooverlap_p$pass NGP_DEBUG_FLAGS=33558528
done
