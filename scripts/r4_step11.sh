#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 900 python scripts/ab_env.py --steps 4 --rounds 2 "" "MCX_DP_LANE_ALWAYS=1" "MCX_DP_LANE_MIN1=65536" "MCX_DP_LANE_MIN1=32768" "MCX_DP_LANE_MIN0=32768 MCX_DP_LANE_MIN1=32768" > gpurun_out/r4_ab9.txt 2> gpurun_out/r4_ab9.err
cut -c1-260 gpurun_out/r4_ab9.txt
