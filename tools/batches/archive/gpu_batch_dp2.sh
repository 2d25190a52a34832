#!/bin/bash
# diagnostics of the data-parallel step's overhead at world size 1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
TAG=${1:-r02dp2}
run() { name=$1; shift
  env "$@" MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout 200 python bench.py --no-cpu-baseline --eval-views 0 --pretrain 1000 --steps 100 --profile-steps 0 $EXTRA > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err
  python -c "import json;d=json.load(open('gpurun_out/${TAG}_$name.json'));print('$name', d['config'].get('dp_backend'), 'ms/step', round(d['ms_per_step'],4), 'rays/s', round(d['value']/1e6,2))"
}
EXTRA="" run single_with_process_group NGP_BENCH_INIT_PG=1
EXTRA="--dp-backend rccl" run dp_rccl_skip_allreduce NGP_FORCE_DP=1 NGP_DP_SKIP_ALLREDUCE=1
EXTRA="--dp-backend rccl" run dp_rccl_opt_flush0 NGP_FORCE_DP=1 AMD_OPT_FLUSH=0
