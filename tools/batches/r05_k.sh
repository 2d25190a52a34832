#!/bin/bash
# round 5, call k: (1) SDF ground truth as ONE persistent launch with lanes refilled from work lists (k_sdf_walks): parity (bit-identical to brute force), then interleaved
# A / B against the round-4 three-launch path (NGP_SDF_PERSISTENT=0); (2) the Adam debias cache: optimizer parity tests; (3) VERDICT r4 item 7: request size of K2-style
# gathers per load flavour (tools/k2_request_size.hip; call j lost 10 minutes to a fault in its scalar flavour under rocprofv3 -- short timeouts now)
R=$PWD; O=gpurun_out/r05k; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests/test_sdf.py -q -x -m gpu > $O/pytest_sdf.log 2>&1; tail -6 $O/pytest_sdf.log | cut -c1-600
for i in 1 2; do
  for v in 1 0; do
    NGP_SDF_PERSISTENT=$v timeout 120 python tools/f4_bench.py sdf > $O/f4_sdf_persistent${v}_$i.jsonl 2> $O/f4_sdf_persistent${v}_$i.err
    echo "persistent=$v run $i"; cut -c1-400 $O/f4_sdf_persistent${v}_$i.jsonl
  done
done
timeout 400 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py -q -x -m gpu -k "optimizer or fused_optimizer or training_loop_tracks" > $O/pytest_adam.log 2>&1; tail -4 $O/pytest_adam.log | cut -c1-600
cd /tmp
timeout 40 $R/tools/k2_request_size 380000 20 > $R/$O/k2_request_size_times.txt 2>&1; cat $R/$O/k2_request_size_times.txt | cut -c1-220
if grep -q "s_load" $R/$O/k2_request_size_times.txt; then
  rm -rf /tmp/pmc_rq
  timeout 60 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d /tmp/pmc_rq -o p -- $R/tools/k2_request_size 380000 5 > $R/$O/pmc_run.log 2>&1
  python $R/tools/rocpd_pmc.py /tmp/pmc_rq/p_results.db k_gather > $R/$O/k2_request_size_pmc.txt 2>&1; cat $R/$O/k2_request_size_pmc.txt | cut -c1-200
  rm -rf /tmp/pmc_tcp
  timeout 60 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d /tmp/pmc_tcp -o p -- $R/tools/k2_request_size 380000 5 >> $R/$O/pmc_run.log 2>&1
  python $R/tools/rocpd_pmc.py /tmp/pmc_tcp/p_results.db k_gather > $R/$O/k2_request_size_pmc_l2.txt 2>&1; tail -70 $R/$O/k2_request_size_pmc_l2.txt | cut -c1-200
fi
