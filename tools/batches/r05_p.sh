#!/bin/bash
# round 5, call p: what bounds the persistent SDF walker -- SQ counters of k_sdf_walks (VALU / LDS / memory wait shares), two passes over tools/f4_bench.py sdf
R=$PWD; O=gpurun_out/r05p; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
for g in a b c; do
  case $g in
    a) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES";;
    b) C="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM";;
    c) C="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE";;
  esac
  rm -rf /tmp/pmc_sdf_$g
  timeout 120 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "k_sdf" -d /tmp/pmc_sdf_$g -o p -- python $R/tools/f4_bench.py sdf > $R/$O/run_$g.log 2>&1
  echo "== group $g: $C" >> $R/$O/pmc_sdf_summary.txt
  python $R/tools/rocpd_pmc.py /tmp/pmc_sdf_$g/p_results.db k_sdf >> $R/$O/pmc_sdf_summary.txt 2>&1
done
cat $R/$O/pmc_sdf_summary.txt | cut -c1-160
