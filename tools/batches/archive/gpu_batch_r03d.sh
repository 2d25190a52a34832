#!/bin/bash
# round 3, call 4: full GPU test tier (new: shapes incl. colour-network depth, lens models on the device, A/B PSNR test), fox pin variants, 3 more A/B seeds at 1k / 5k
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=r03d
date
timeout 1200 python -m pytest tests/test_gpu_shapes.py tests/test_gpu_lens.py tests/test_gpu_ab_psnr.py -m gpu -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest_new.log 2>&1; echo "new tests rc $?"
grep -E "passed|failed|FAILED|held-out PSNR|rays, max|coverage" gpurun_out/${TAG}_pytest_new.log | cut -c1-330 | tail -40
date
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_shapes.py --deselect tests/test_gpu_lens.py --deselect tests/test_gpu_ab_psnr.py > gpurun_out/${TAG}_pytest_rest.log 2>&1; echo "rest rc $?"
tail -6 gpurun_out/${TAG}_pytest_rest.log | cut -c1-400
date
timeout 600 python tools/fox_notebook_pin.py 2000 gpurun_out/${TAG}_fox_notebook_pin.json > gpurun_out/${TAG}_fox_pin.log 2>&1; echo "fox pin rc $?"
python -c "import json;d=json.load(open('gpurun_out/${TAG}_fox_notebook_pin.json'));print({k:(round(v['loss_tail_mean'],6),round(v['loss_tail_std'],6),round(v['ratio_tail_mean_to_notebook'],3)) for k,v in d.items()})"
date
timeout 600 python bench.py --pretrain 200 --steps 20 --warmup 5 --no-cpu-baseline --no-fox-leg --eval-views 8 --eval-res 800 --eval-spp 8 --ab-psnr 1000,5000 --ab-seeds 3 --ab-seed0 1342 --profile-steps 4 > gpurun_out/${TAG}_bench_ab3.json 2> gpurun_out/${TAG}_bench_ab3.err; echo "ab rc $?"
python -c "import json;d=json.load(open('gpurun_out/${TAG}_bench_ab3.json'));print(json.dumps(d['config'].get('ab_psnr')))" | cut -c1-2000
date
