#!/bin/bash
# round 6 (second session), call bn: K3 is bound by its span atomics (one returning atomic per workgroup on ONE counter word, ~90-100 per us: 3.6 k workgroups of 16 rays = 38 us):
# 16-wavefront workgroups (half the atomics) at 64 registers / 80 B scratch = two per CU (NGP_K3_VARIANT=1), 8-wavefront workgroups at 64 registers = four per CU (=2), production (=0)
R=$PWD; O=gpurun_out/r06bn; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
. tools/batches/ab_lib.sh
for pass in 1 2 3; do for v in 0 1 2; do ab_run k3v${v}_p$pass NGP_K3_VARIANT=$v; done; done
