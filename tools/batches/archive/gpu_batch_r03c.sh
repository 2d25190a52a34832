#!/bin/bash
# round 3, call 3: helper streams created before / after the communicator; the notebook's fox number; 5-seed A/B PSNR
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03c
date
: > gpurun_out/${TAG}_dp_diag.jsonl
run() { name=$1; comm=$2; shift 2; env "$@" timeout 300 python tools/dp_diag.py $name $comm 2>gpurun_out/${TAG}_dp_diag_$name.err | tail -1 | tee -a gpurun_out/${TAG}_dp_diag.jsonl | cut -c1-420; }
run single nocomm NGP_X=1
run comm_split_eager_streams comm NGP_X=1
run comm_split_lazy_streams comm NGP_LAZY_STREAMS=1
run comm_fused_eager_streams comm NGP_DP_FUSED_STEP=1
date
NGP_FORCE_DP=1 timeout 300 python bench.py --no-cpu-baseline --eval-views 0 --profile-steps 4 > gpurun_out/${TAG}_bench_forced_dp.json 2> gpurun_out/${TAG}_bench_forced_dp.err; echo "forced dp rc $?"; cut -c1-420 gpurun_out/${TAG}_bench_forced_dp.json
date
timeout 600 python tools/fox_notebook_pin.py 2000 gpurun_out/${TAG}_fox_notebook_pin.json > gpurun_out/${TAG}_fox_pin.log 2>&1; echo "fox pin rc $?"; tail -c 1500 gpurun_out/${TAG}_fox_pin.log
date
timeout 1500 python bench.py --pretrain 200 --steps 20 --warmup 5 --no-cpu-baseline --eval-views 8 --eval-res 800 --eval-spp 8 --ab-psnr 1000,5000,20000 --ab-seeds 5 --profile-steps 4 > gpurun_out/${TAG}_bench_ab5.json 2> gpurun_out/${TAG}_bench_ab5.err; echo "ab rc $?"
python -c "import json;d=json.load(open('gpurun_out/${TAG}_bench_ab5.json'));print(json.dumps(d['config'].get('ab_psnr')))" | cut -c1-3000
date
