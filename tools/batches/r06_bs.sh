#!/bin/bash
# round 6 (second session), call bs: the evaluation renderer (run.py's PSNR procedure: 8 views 800 x 800, spp 8) -- time per frame and per-kernel table; last measured in round 2 (3.05 ms per frame)
R=$PWD; O=$R/gpurun_out/r06bs; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python tools/render_bench.py 1500 8 800 8 2>&1 | tail -1 | tee $O/render_bench.log
cd /tmp && rm -rf /tmp/prof_r && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o t -- python $R/tools/render_bench.py 1500 8 800 8 > $O/rocprof.log 2>&1; echo "rocprof rc $?"
cd $R; find /tmp/prof_r -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_render.csv \;
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r06bs/kernel_stats_render.csv")))
for r in rows[:14]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg us {float(r["AverageNs"])/1e3:9.1f} total ms {float(r["TotalDurationNs"])/1e6:8.1f}')
PY
