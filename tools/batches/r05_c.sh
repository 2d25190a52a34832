#!/bin/bash
# round 5, call c: what bounds k_train_fused (51 us for ~1,600 issue slots x 8 tiles per wave pair): SQ wait / active / MFMA counters
R=$PWD; O=gpurun_out/r05c; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python tools/pmc_probe.py $R/$O/pmc 300 6 default sq,sq2,mfma > $O/pmc.log 2>&1
grep -E "group|k_train_fused|k_grad_bin|k_grad_accumulate|k_inference_tiles" $O/pmc_summary.txt | cut -c1-900
