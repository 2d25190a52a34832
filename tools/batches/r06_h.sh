#!/bin/bash
# round 6, call h: the rest of the tests behind the EMA change (r06_g stopped at the wire-format test, whose expectation was the round-5 layout)
R=$PWD; O=gpurun_out/r06h; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_pyngp.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "$(tail -4 $O/pytest.log | cut -c1-300)"
