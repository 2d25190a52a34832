#!/bin/bash
# HBM-side traffic of the dominant kernels from PMC counters (GPU box). FETCH_SIZE and WRITE_SIZE need separate passes
# (TCC slot limits, MI355X_MICROARCH.md); the rocpd databases are summarised on the box and deleted (they are large).
# usage: tools/pmc_traffic.sh <out_prefix>
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/pmc}
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 400 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "k_train_fused|k_train_fwd_bwd|k_grad_bin|k_grad_accumulate|k_inference|k_optimizer|k_wgrad|k_compute_loss_v2|k1_count|k1_write" -d /tmp/pmc_$C -o p -- python $R/tools/microbench.py 1000 8 default > ${OUT}_$C.log 2>&1
  python $R/tools/rocpd_pmc.py /tmp/pmc_$C/p_results.db > ${OUT}_$C.txt 2>&1
  rm -rf /tmp/pmc_$C
done
tail -n 40 ${OUT}_FETCH_SIZE.txt ${OUT}_WRITE_SIZE.txt
