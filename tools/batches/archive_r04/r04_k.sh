#!/bin/bash
R=$PWD; O=gpurun_out/r04k; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python tools/pmc_probe.py $R/$O/pmc 1000 8 default tcc,rdsize,tcp > $O/pmc_probe.log 2>&1; echo rc $?
grep -E "group|k_encode_tiles_xcd|k_inference_tiles" $O/pmc_summary.txt | cut -c1-400
