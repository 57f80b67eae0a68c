#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 1200 python -m pytest tests -m gpu -q -x -k "sam_equals or full_size or large_batch or config5 or smoke or degenerate or long_cigars" -p no:cacheprovider 2>&1 | tail -4
timeout 900 python scripts/ab_env.py --steps 3 "" "MCX_RESCUE_IN_LINE=1" "MCX_LATE_RESEED=1" "MCX_TIER0_CAPS=88,24,128,2048" "MCX_TIER0_CAPS=64,16,96,2048" "" > gpurun_out/r4_ab5_cfg3.txt 2> gpurun_out/r4_ab5_cfg3.err
cat gpurun_out/r4_ab5_cfg3.txt
for w in w4 w3; do MCX_LIB=$PWD/mapcaller_amd/libmcx_$w.so timeout 600 python scripts/ab_env.py --steps 3 "" "" 2>/dev/null | sed "s/^/$w /"; done
export MCX_TIMING=1
TAG=cfg3e bash scripts/timeline.sh > gpurun_out/r4_timeline_cfg3e.txt 2>&1
grep -E "after clustering|pairs 4000000" gpurun_out/timeline_cfg3e/kt.log | tail -3
head -64 gpurun_out/timeline_cfg3e/timeline.txt
