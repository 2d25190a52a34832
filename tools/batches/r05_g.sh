#!/bin/bash
# round 5, call g: A/B PSNR production vs reference order, 12 seeds on fox (2 k / 5 k steps, 16 training views at full resolution, spp 2), four trainings side by side
R=$PWD; O=gpurun_out/r05g; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python tools/ab_psnr_parallel.py $R/$O/ab_psnr_fox_12seeds.json fox 2000,5000 12 4 --eval-views 16 --eval-spp 2 > $O/ab_fox.log 2>&1; tail -5 $O/ab_fox.log | cut -c1-1500
