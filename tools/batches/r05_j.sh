#!/bin/bash
# round 5, call j: VERDICT r4 item 7 -- request size of K2-style gathers per load flavour (stand-alone model, tools/k2_request_size.hip): timings, then one PMC pass (read requests by size)
R=$PWD; O=gpurun_out/r05j; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 120 $R/tools/k2_request_size 380000 20 > $R/$O/k2_request_size_times.txt 2>&1; cat $R/$O/k2_request_size_times.txt | cut -c1-220
rm -rf /tmp/pmc_rq
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d /tmp/pmc_rq -o p -- $R/tools/k2_request_size 380000 5 > $R/$O/pmc_run.log 2>&1
python $R/tools/rocpd_pmc.py /tmp/pmc_rq/p_results.db k_gather > $R/$O/k2_request_size_pmc.txt 2>&1; cat $R/$O/k2_request_size_pmc.txt | cut -c1-200
rm -rf /tmp/pmc_tcp
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d /tmp/pmc_tcp -o p -- $R/tools/k2_request_size 380000 5 >> $R/$O/pmc_run.log 2>&1
python $R/tools/rocpd_pmc.py /tmp/pmc_tcp/p_results.db k_gather > $R/$O/k2_request_size_pmc_l2.txt 2>&1; tail -60 $R/$O/k2_request_size_pmc_l2.txt | cut -c1-200
