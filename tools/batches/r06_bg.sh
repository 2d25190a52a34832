#!/bin/bash
# round 6 (second session), call bg: SQ counters of the fox leg's K1 kernels (k1_count<8, false> is 125 us for ~4.3 k rays: what does a ray's chain consist of?)
R=$PWD; O=$R/gpurun_out/r06bg; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
for g in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "sq2 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  set -- $g; name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "k1_count|k1_write|k1_setup" -d /tmp/pmc_$name -o p -- python $R/bench.py --gpus 1 --scene fox --pretrain 600 --steps 30 --warmup 5 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --no-calibration --eval-views 0 --steady-steps 0 > $O/pmc_$name.log 2>&1; echo "pmc $name rc $?"
  echo "== group $name: $@" >> $O/pmc_fox_k1_summary.txt
  python $R/tools/rocpd_pmc.py /tmp/pmc_$name/p_results.db >> $O/pmc_fox_k1_summary.txt 2>&1
  rm -rf /tmp/pmc_$name
done
cat $O/pmc_fox_k1_summary.txt | cut -c1-260
