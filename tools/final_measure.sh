#!/bin/bash
# Round-end measurement batch on the GPU box: smoke, tests, bench (full JSON line), rocprofv3 kernel stats of the bench command,
# PMC traffic passes.  usage: tools/final_measure.sh <tag>
set -u
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r01}
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1 | tee gpurun_out/${TAG}_smoke.log
timeout 800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tee gpurun_out/${TAG}_pytest_gpu.log | tail -4
timeout 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -2 gpurun_out/${TAG}_bench.err; cut -c1-700 gpurun_out/${TAG}_bench.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 NGP_FORCE_DP=1 timeout 200 python bench.py --no-cpu-baseline --eval-views 0 --pretrain 300 --steps 50 > gpurun_out/${TAG}_bench_forced_dp1.json 2> gpurun_out/${TAG}_bench_forced_dp1.err; cut -c1-300 gpurun_out/${TAG}_bench_forced_dp1.json; tail -1 gpurun_out/${TAG}_bench_forced_dp1.err
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_bench
# stream overlap off (NGP_DEBUG_FLAGS=4096) so that the per-kernel durations are those of isolated kernels, like the HIP-event leg of bench.py
NGP_DEBUG_FLAGS=4096 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $R/bench.py --no-cpu-baseline --eval-views 0 > $R/gpurun_out/${TAG}_rocprof_bench.log 2>&1
for f in $(find /tmp/prof_bench -name "*kernel_stats.csv" -o -name "*domain_stats.csv"); do cp $f $R/gpurun_out/${TAG}_$(basename $f); done
rm -rf /tmp/prof_bench
$R/tools/pmc_traffic.sh $R/gpurun_out/${TAG}_pmc > /dev/null 2>&1; grep -A3 "k_train_fwd_bwd\|k_grad_bin\|k_grad_accumulate\|k_inference_tiles" $R/gpurun_out/${TAG}_pmc_FETCH_SIZE.txt $R/gpurun_out/${TAG}_pmc_WRITE_SIZE.txt | cut -c1-170 | head -40
