#!/bin/bash
# round 5, call i: the whole GPU tier at HEAD, with twelve more A/B seeds of the lego-format scene (latency-bound reference-order trainings) running beside it
R=$PWD; O=gpurun_out/r05i; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
AB_SEED0=1349 timeout 1700 python tools/ab_psnr_parallel.py $R/$O/ab_psnr_synthetic_12seeds_b.json synthetic 2000,5000 12 4 --eval-views 8 --eval-res 400 --eval-spp 4 > $O/ab_syn_b.log 2>&1 &
P1=$!
timeout 1700 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -15 $O/pytest_gpu.log | cut -c1-600
wait $P1
tail -2 $O/ab_syn_b.log | cut -c1-900
