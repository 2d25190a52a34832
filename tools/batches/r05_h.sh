#!/bin/bash
# round 5, call h: A/B PSNR, twelve more fox seeds (1349..1360) and twelve seeds on the lego-format scene (8 held-out views 400^2, spp 4), both experiments side by side
R=$PWD; O=gpurun_out/r05h; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
AB_SEED0=1349 timeout 1700 python tools/ab_psnr_parallel.py $R/$O/ab_psnr_fox_12seeds_b.json fox 2000,5000 12 4 --eval-views 16 --eval-spp 2 > $O/ab_fox_b.log 2>&1 &
P1=$!
timeout 1700 python tools/ab_psnr_parallel.py $R/$O/ab_psnr_synthetic_12seeds.json synthetic 2000,5000 12 4 --eval-views 8 --eval-res 400 --eval-spp 4 > $O/ab_syn.log 2>&1 &
P2=$!
wait $P1; wait $P2
tail -3 $O/ab_fox_b.log | cut -c1-1200; tail -3 $O/ab_syn.log | cut -c1-1200
