#!/bin/bash
# Round-2 GPU batch F: measurement deliverables -- headline bench line, full-scale A/B PSNR, fox line, calibrated traffic counters
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02f}
echo "== bench (headline)"; date
timeout 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc $?"; cut -c1-900 gpurun_out/${TAG}_bench.json
echo "== pmc request sizes"; date
timeout 600 python tools/pmc_probe.py $R/gpurun_out/${TAG}_pmc 1000 8 default rdsize,wrsize > gpurun_out/${TAG}_pmc_probe.log 2>&1
python tools/pmc_traffic_json.py gpurun_out/${TAG}_pmc_summary.txt gpurun_out/${TAG}_pmc_traffic.json
echo "== bench fox"; date
timeout 600 python bench.py --scene fox --pretrain 5000 --steps 200 --warmup 20 --eval-views 4 --no-cpu-baseline > gpurun_out/${TAG}_bench_fox.json 2> gpurun_out/${TAG}_bench_fox.err; echo "fox rc $?"
cut -c1-1200 gpurun_out/${TAG}_bench_fox.json
echo "== A/B psnr full scale"; date
timeout 1500 python bench.py --pretrain 200 --steps 20 --warmup 5 --no-cpu-baseline --eval-views 8 --eval-res 800 --eval-spp 8 --ab-psnr 1000,5000,20000 --profile-steps 4 > gpurun_out/${TAG}_bench_ab.json 2> gpurun_out/${TAG}_bench_ab.err; echo "ab rc $?"
python -c "import json;d=json.load(open('gpurun_out/${TAG}_bench_ab.json'));print(json.dumps(d['config'].get('ab_psnr')))"
date
