#!/bin/bash
# round 6, call y: A/B PSNR production vs reference order on the lego-HARD stand-in at 5 k / 20 k steps, 12 seeds, 8 held-out views 800^2 spp 4 (with every kernel change of the round in the library)
R=$PWD; O=gpurun_out/r06y; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 3300 python tools/ab_psnr_parallel.py $R/$O/ab_psnr_hard_12seeds.json hard 5000,20000 12 4 --eval-views 8 --eval-res 800 --eval-spp 4 --psnr-steps "" > $O/ab_hard.log 2>&1; tail -4 $O/ab_hard.log | cut -c1-1500
