#!/bin/bash
# round 6, call t: the two Appendix-A switches in numbers -- (1) EMA half- vs full-precision state (tools/ab_ema.py, deterministic K3: identical trainings, the test PSNR differs through the
# inference weights only), synthetic + fox, 3 seeds, 2 k / 10 k steps; (2) fp32 vs half matrix-multiply accumulators in the ORACLE (tools/ab_half_accumulate.py, CPU side on the box's host cores)
R=$PWD; O=gpurun_out/r06t; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python tools/ab_ema.py synthetic 2000,10000 3 > $O/ab_ema_synthetic.json 2> $O/ab_ema_synthetic.err; echo "ema synthetic rc $?"; cut -c1-900 $O/ab_ema_synthetic.json
timeout 900 python tools/ab_ema.py fox 2000,10000 3 > $O/ab_ema_fox.json 2> $O/ab_ema_fox.err; echo "ema fox rc $?"; cut -c1-900 $O/ab_ema_fox.json
timeout 1200 python tools/ab_half_accumulate.py --steps 1500 --cpu-steps 24 --res 96 > $O/ab_half_accumulate.json 2> $O/ab_half_accumulate.err; echo "half-acc rc $?"; cut -c1-1500 $O/ab_half_accumulate.json; tail -3 $O/ab_half_accumulate.err
