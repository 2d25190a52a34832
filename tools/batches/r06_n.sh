#!/bin/bash
# round 6, call n: A/B PSNR production vs reference order at 10 k / 20 k steps on fox, 20 seeds (the point VERDICT r4 1(b) / r5 7(d) still miss), 16 training views at full resolution, spp 2,
# four trainings side by side
R=$PWD; O=gpurun_out/r06n; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 3300 python tools/ab_psnr_parallel.py $R/$O/ab_psnr_fox_20k_20seeds.json fox 10000,20000 20 4 --eval-views 16 --eval-spp 2 --psnr-steps "" > $O/ab_fox.log 2>&1; tail -5 $O/ab_fox.log | cut -c1-1500
