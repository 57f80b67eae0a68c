#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
SECONDS=0
timeout 2700 python -m pytest tests -m gpu -q --timeout 2400 -p no:cacheprovider > gpurun_out/r5_full_test.log 2>&1
echo "tests: $SECONDS s" >> gpurun_out/r5_full_test.log; tail -6 gpurun_out/r5_full_test.log | cut -c1-300
