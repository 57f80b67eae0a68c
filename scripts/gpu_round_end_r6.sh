#!/bin/bash
# round end: the whole GPU suite with its slowest tests, the .gz rates, the smoke entry (gpurun -- bash scripts/gpu_round_end_r6.sh)
mkdir -p gpurun_out/r6k; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
( time timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/r6k/gpu_suite.log 2>&1 < /dev/null
tail -26 gpurun_out/r6k/gpu_suite.log
timeout 900 python3 scripts/gz_rate.py > gpurun_out/r6k/gz_rate.json 2> gpurun_out/r6k/gz_rate.err < /dev/null
cat gpurun_out/r6k/gz_rate.json
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6k/smoke.log 2>&1; tail -2 gpurun_out/r6k/smoke.log
