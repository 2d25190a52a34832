#!/bin/bash
# round 6 (second session), call bt: equal-step PSNR production - reference order on FOX at 2 k / 5 k steps, 12 seeds, with the final code (the two-phase multi-cascade marcher, the grid
# samples drawn ahead): the round-5 24-seed intervals were taken with the old k1_count
R=$PWD; O=gpurun_out/r06bt; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 2400 python tools/ab_psnr_parallel.py $R/$O/ab_psnr_fox_2k_5k_12seeds.json fox 2000,5000 12 4 --eval-views 16 --eval-spp 2 --psnr-steps "" > $O/ab_fox.log 2>&1; tail -4 $O/ab_fox.log | cut -c1-1500
