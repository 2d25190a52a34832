#!/bin/bash
# Round-6 measurement batch (gpu_batch_r03_final.sh with the driver's bench command): smoke, full GPU tests, calibrated traffic counters, headline bench line (reads the fresh traffic file),
# kernel traces of the bench command with and without stream overlap, fox line, A/B PSNR, ablation table.  usage: tools/gpu_batch_final.sh <tag> [parts]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=${1:-r06final}
PARTS=${2:-smoke,pytest,pmc,bench,trace}
has() { case ",$PARTS," in *,$1,*) return 0;; *) return 1;; esac; }
if has smoke; then echo "== smoke"; date; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.log; fi
if has pytest; then echo "== pytest"; date
  timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?"
  grep -E "passed|failed" gpurun_out/${TAG}_pytest_gpu.log | tail -3; fi
if has pmc; then echo "== pmc request sizes + mfma"; date
  timeout 900 python tools/pmc_probe.py $R/gpurun_out/${TAG}_pmc 1000 8 default rdsize,wrsize,mfma > gpurun_out/${TAG}_pmc_probe.log 2>&1; echo "pmc rc $?"
  python tools/pmc_traffic_json.py gpurun_out/${TAG}_pmc_summary.txt gpurun_out/${TAG}_pmc_traffic.json && cp gpurun_out/${TAG}_pmc_traffic.json profiles/r06_pmc_traffic.json
  python -c "import json;d=json.load(open('gpurun_out/${TAG}_pmc_traffic.json'));print({k:{a:round(b/1e6,1) for a,b in v.items() if a in ('read_bytes','write_bytes')} for k,v in d.items() if isinstance(v,dict)})" | cut -c1-900; fi
if has bench; then echo "== bench (headline, the driver's command)"; date
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc $?"; cut -c1-1500 gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err | cut -c1-300; fi
prof() { # tag, env...
  tag=$1; shift
  cd /tmp && rm -rf /tmp/prof_$tag && env "$@" timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o t -- python $R/bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --no-calibration --eval-views 0 --steady-steps 0 > $R/gpurun_out/${TAG}_rocprof_$tag.log 2>&1; echo "rocprof $tag rc $?"
  cd $R
  find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats_$tag.csv \;
  T=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
  python tools/kernel_trace_summary.py "$T" > gpurun_out/${TAG}_kernel_trace_summary_$tag.txt 2>&1
  grep -A16 "average step timeline" gpurun_out/${TAG}_kernel_trace_summary_$tag.txt | cut -c1-130
  grep "steady-state" gpurun_out/${TAG}_kernel_trace_summary_$tag.txt | cut -c1-400
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/${TAG}_rocprof_$tag.log | head -2
  rm -rf /tmp/prof_$tag
}
if has trace; then echo "== rocprofv3 kernel traces of the bench command"; date
  prof nooverlap NGP_DEBUG_FLAGS=4096
  prof overlap NGP_X=1; fi
if has fox; then echo "== bench fox"; date
  timeout 600 python bench.py --scene fox --pretrain 5000 --steps 200 --warmup 20 --eval-views 4 --no-cpu-baseline --no-fox-leg > gpurun_out/${TAG}_bench_fox.json 2> gpurun_out/${TAG}_bench_fox.err; echo "fox rc $?"
  cut -c1-1200 gpurun_out/${TAG}_bench_fox.json; fi
if has dpsum; then echo "== fp16 vs fp32 sum of 8 shard gradients"; date
  timeout 600 python tools/dp_fp16_sum_error.py 8 1000 > gpurun_out/${TAG}_dp_fp16_sum_error.json 2> gpurun_out/${TAG}_dp_fp16_sum_error.err; echo "dpsum rc $?"; cut -c1-1200 gpurun_out/${TAG}_dp_fp16_sum_error.json; fi
if has ab; then echo "== A/B psnr full scale"; date
  timeout 1500 python bench.py --pretrain 200 --steps 20 --warmup 5 --no-cpu-baseline --eval-views 8 --eval-res 800 --eval-spp 8 --ab-psnr 1000,5000,20000 --profile-steps 4 > gpurun_out/${TAG}_bench_ab.json 2> gpurun_out/${TAG}_bench_ab.err; echo "ab rc $?"
  python -c "import json;d=json.load(open('gpurun_out/${TAG}_bench_ab.json'));print(json.dumps(d['config'].get('ab_psnr')))"; fi
if has ablation; then echo "== ablation table"; date
  timeout 600 python tools/microbench.py 1000 32 default,k3_one_ray_per_wave,k1_no_first_point_skip,k2_tile32,k2_rounds3,grid_no_sort,k1_no_prefilter,w_single_role,t1_dense_atomics,bin_no_hashed_merge,bin_chunk12_split,k2_eager,t1_no_binning,default_again > gpurun_out/${TAG}_microbench.log 2> gpurun_out/${TAG}_microbench.err; echo "microbench rc $?"
  cut -c1-700 gpurun_out/${TAG}_microbench.log; fi
date
