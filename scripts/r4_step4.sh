#!/bin/bash
# round 4, fourth run: in-lane DP in k_simple, adaptive lane/wave DP, fewer streams — parity, A/B, timeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -q -x -k "sam_equals or extend or degenerate or long_cigars or cli_sam or overlapped or packed_host or full_size or large_batch or config5 or config2 or smoke or fuzz_rounds_equal" -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/r4_step4_pytest.log
tail -5 gpurun_out/r4_step4_pytest.log
timeout 900 python scripts/ab_env.py --steps 3 "" "MCX_NO_SIMPLE=1" "MCX_DP_LANE_ALWAYS=1" "" > gpurun_out/r4_ab2_cfg3.txt 2> gpurun_out/r4_ab2_cfg3.err
cat gpurun_out/r4_ab2_cfg3.txt
export MCX_TIMING=1
TAG=cfg3b bash scripts/timeline.sh > gpurun_out/r4_timeline_cfg3b.txt 2>&1
grep -E "after clustering|pairs 4000000" gpurun_out/timeline_cfg3b/kt.log | tail -4
head -70 gpurun_out/timeline_cfg3b/timeline.txt
