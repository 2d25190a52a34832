#!/bin/bash
R=$PWD; O=gpurun_out/r04f; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -m pytest tests/test_gpu_model.py tests/test_gpu_dist.py tests/test_gpu_train.py -q -s -k "training_step_gradients or tracks_oracle or fused_optimizer or multi_rank" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
