#!/bin/bash
# the GPU suite and a short default bench on the current build
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 3000 python -m pytest tests -m gpu -q --timeout 2400 -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r4_pytest_final.log; tail -4 gpurun_out/r4_pytest_final.log
python bench.py --steps 10 --warmup 2 --other-configs 0 --second-genome 0 --cpu-pairs 0 --file-steps 0 --pcie-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms_per_step'])"
