#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 900 python -m pytest tests -m gpu -q -x -k "sam_equals or full_size or large_batch or config5 or smoke" -p no:cacheprovider 2>&1 | tail -4
timeout 900 python scripts/ab_env.py --steps 3 "" "" > gpurun_out/r4_ab3_cfg3.txt 2> gpurun_out/r4_ab3_cfg3.err
cat gpurun_out/r4_ab3_cfg3.txt
export MCX_TIMING=1
TAG=cfg3c bash scripts/timeline.sh > gpurun_out/r4_timeline_cfg3c.txt 2>&1
grep -E "after clustering|pairs 4000000" gpurun_out/timeline_cfg3c/kt.log | tail -3
head -64 gpurun_out/timeline_cfg3c/timeline.txt
