#!/bin/bash
# single rank vs forced data-parallel step (communicator of one rank) at the trained steady state, both all-reduce backends
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
TAG=${1:-r02dp}
run() { name=$1; shift
  env "$@" MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout 200 python bench.py --no-cpu-baseline --eval-views 0 --pretrain 1000 --steps 100 --profile-steps 0 $EXTRA > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err
  python -c "import json;d=json.load(open('gpurun_out/${TAG}_$name.json'));print('$name', d['config'].get('dp_backend'), 'ms/step', round(d['ms_per_step'],4), 'rays/s', round(d['value']/1e6,2), d['config'].get('dp_host_enqueue_ms_per_step'))"
}
EXTRA="" run single A=0
EXTRA="--dp-backend rccl" run dp_rccl NGP_FORCE_DP=1
EXTRA="--dp-backend torch" run dp_torch NGP_FORCE_DP=1
EXTRA="" run single_again A=0
