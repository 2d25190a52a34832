#!/bin/bash
# round 6 (second session), call ba: is the SDF ground-truth launch as long as its longest chain?  G batches' BVH halves in one launch (tools/exp_sdf_multibatch.py), 2 and 4 workgroups per CU
R=$PWD; O=gpurun_out/r06ba; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for occ in 2 4 2; do
  NGP_SDF_WALK_OCC=$occ timeout 200 python tools/exp_sdf_multibatch.py >> $O/sdf_multibatch.jsonl 2>> $O/sdf_multibatch.err; echo "occ $occ rc $?"
done
cat $O/sdf_multibatch.jsonl; tail -3 $O/sdf_multibatch.err
