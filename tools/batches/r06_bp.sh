#!/bin/bash
# round 6 (second session), call bp: the multi-rank code path of bench.py at world size 1 with the final code: (1) the driver's N > 1 launch line with N = 1 under torch.distributed.run,
# (2) NGP_FORCE_DP=1 (process group + in-library RCCL communicator of one rank + the sharded step) -- sanity before a node with more than one GPU ever runs it
R=$PWD; O=gpurun_out/r06bp; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --eval-views 0 > $O/torchrun_n1.json 2> $O/torchrun_n1.err; echo "torchrun rc $?"
NGP_FORCE_DP=1 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --eval-views 0 > $O/forced_dp.json 2> $O/forced_dp.err; echo "forced dp rc $?"; tail -2 $O/forced_dp.err | cut -c1-300
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --no-hard-leg --no-f4-legs --eval-views 0 > $O/plain.json 2> $O/plain.err; echo "plain rc $?"
python - <<'PY'
import json
for n in ("torchrun_n1", "forced_dp", "plain"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r06bp/{n}.json") if l.startswith("{")][-1])
        print(n, round(d["value"] / 1e6, 2), "M rays/s", round(d["ms_per_step"], 4), "ms", d["config"].get("parallelism"), d["config"].get("dp_backend"), "n_gpus", d["n_gpus"])
    except Exception as e:
        print(n, "FAILED", e)
PY
