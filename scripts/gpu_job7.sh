#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 1700 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider --durations=12 2>&1 | tail -40 > gpurun_out/r2_pytest9.log; tail -28 gpurun_out/r2_pytest9.log | cut -c1-300
