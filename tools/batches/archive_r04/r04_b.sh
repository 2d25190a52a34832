#!/bin/bash
# round 4, call b: GPU tier at HEAD + fox A/B with both clamp variants
mkdir -p gpurun_out/r04b
python -m pytest tests -m gpu -q -s > gpurun_out/r04b/pytest.log 2>&1
tail -5 gpurun_out/r04b/pytest.log
python bench.py --scene fox --steps 20 --warmup 5 --pretrain 300 --no-cpu-baseline --ab-psnr 2000,5000 --ab-seeds 5 --ab-clamp-variants > gpurun_out/r04b/ab_fox.log 2>&1
tail -c 3000 gpurun_out/r04b/ab_fox.log
