#!/bin/bash
# Round-2 GPU batch G: tests (Rfl modes, ray scramble), A/B PSNR again, render timing trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02g}
timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?"
grep -v "^$" gpurun_out/${TAG}_pytest_gpu.log | grep "passed\|failed\|FAILED\|Error\|train_mode" | head -20 | cut -c1-300
echo "== A/B psnr full scale"; date
timeout 1500 python bench.py --pretrain 200 --steps 20 --warmup 5 --no-cpu-baseline --eval-views 8 --eval-res 800 --eval-spp 8 --ab-psnr 1000,5000,20000 --profile-steps 4 > gpurun_out/${TAG}_bench_ab.json 2> gpurun_out/${TAG}_bench_ab.err; echo "ab rc $?"
python -c "import json;d=json.load(open('gpurun_out/${TAG}_bench_ab.json'));print(json.dumps(d['config'].get('ab_psnr')))"
echo "== render trace"; date
cd /tmp && rm -rf /tmp/prof_r && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o t -- python $R/bench.py --pretrain 1500 --steps 10 --warmup 2 --no-cpu-baseline --eval-views 8 --eval-res 800 --eval-spp 2 --profile-steps 0 > $R/gpurun_out/${TAG}_rocprof_render.log 2>&1; echo "rocprof rc $?"
cd $R
T=$(find /tmp/prof_r -name "*kernel_trace.csv" | head -1)
python tools/kernel_trace_summary.py "$T" > gpurun_out/${TAG}_kernel_trace_summary_render.txt 2>&1
grep "k_render\|k_inference<false\|dispatches\|steady" gpurun_out/${TAG}_kernel_trace_summary_render.txt | cut -c1-200
date
