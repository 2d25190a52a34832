#!/bin/bash
# round 6, call u: fp32 vs half matrix-multiply accumulators in the ORACLE (tools/ab_half_accumulate.py; r06_t's attempt passed a null depth buffer to the oracle's renderer)
R=$PWD; O=gpurun_out/r06u; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python tools/ab_half_accumulate.py --steps 1500 --cpu-steps 24 --res 96 > $O/ab_half_accumulate.json 2> $O/ab_half_accumulate.err; echo "half-acc rc $?"; cut -c1-1800 $O/ab_half_accumulate.json; tail -3 $O/ab_half_accumulate.err
