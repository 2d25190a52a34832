#!/bin/bash
# round 6, call j: the driver's command with the new default line (auto scene, timed-window bookkeeping, 200-step steady window, PSNR@{1k,5k,10k,35k} on the headline scene, fox and the hard stand-in)
R=$PWD; O=gpurun_out/r06j; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
T0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $? wall $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06j/bench.json').read().strip().splitlines()[-1])
print(d['metric']); print('value', d['value'], 'ms', d['ms_per_step'])
c=d['config']; print(c['timed_window'], c.get('steady_window')); print('curve', c.get('test_psnr_curve_db'), c.get('test_psnr_curve_seconds'))
for k,v in d.get('legs',{}).items(): print(k, {a:b for a,b in v.items() if a not in ('workload','split_note')})
print(d['cpu_baseline']['value'], d['roofline']['kernel_ms_per_step'])
PY
tail -3 $O/bench.err | cut -c1-300
