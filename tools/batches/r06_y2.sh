#!/bin/bash
# round 6, call y2: twelve more seeds (1349..1360) of the hard-scene A/B (r06_y: the 5 k interval was [-0.08, 0.13] with one +0.47 dB pair among 12); merged with tools/ab_psnr_parallel.py --merge
R=$PWD; O=gpurun_out/r06y2; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
AB_SEED0=1349 timeout 3300 python tools/ab_psnr_parallel.py $R/$O/ab_psnr_hard_12seeds_b.json hard 5000,20000 12 4 --eval-views 8 --eval-res 800 --eval-spp 4 --psnr-steps "" > $O/ab_hard.log 2>&1; tail -2 $O/ab_hard.log | cut -c1-1200
