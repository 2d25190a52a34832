#!/bin/bash
# round 6, call m: per-kernel table of the image and SDF training steps (tools/f4_bench.py under rocprofv3 --kernel-trace --stats)
R=$PWD; O=gpurun_out/r06m; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for what in image sdf; do
cd /tmp && rm -rf /tmp/prof_$what && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$what -o t -- python $R/tools/f4_bench.py $what > $R/$O/rocprof_$what.log 2>&1; echo "rocprof $what rc $?"
cd $R
S=$(find /tmp/prof_$what -name "*kernel_stats.csv" | head -1); cp $S $O/kernel_stats_$what.csv
python - $O/kernel_stats_$what.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]: print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"])/1e3:8.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms  {float(r["Percentage"]):5.1f} %')
PY
done
