#!/bin/bash
# Round-2 GPU batch I: render timing + trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02i}
timeout 300 python tools/render_bench.py 1500 8 800 8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_render_bench.log
cd /tmp && rm -rf /tmp/prof_r && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o t -- python $R/tools/render_bench.py 1500 4 800 2 > $R/gpurun_out/${TAG}_rocprof_render.log 2>&1; echo "rocprof rc $?"
cd $R
T=$(find /tmp/prof_r -name "*kernel_trace.csv" | head -1)
python tools/kernel_trace_summary.py "$T" > gpurun_out/${TAG}_kernel_trace_summary_render.txt 2>&1
grep "k_render\|k_inference<false\|dispatches\|copyBuffer\|fillBuffer" gpurun_out/${TAG}_kernel_trace_summary_render.txt | cut -c1-130
grep "eval " gpurun_out/${TAG}_rocprof_render.log
