#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export MCX_TIMING=1
TAG=cfg5 BENCH_ARGS="--rlen 250 --ins 0.025 --dele 0.025 --alg nw" bash scripts/timeline.sh > gpurun_out/r4_timeline_cfg5.txt 2>&1
grep -E "after clustering|^\[tier 1\]|pairs 4000000" gpurun_out/timeline_cfg5/kt.log | tail -12
TAG=cfg3 bash scripts/timeline.sh > gpurun_out/r4_timeline_cfg3.txt 2>&1
grep -E "after clustering|pairs 4000000" gpurun_out/timeline_cfg3/kt.log | tail -6
head -60 gpurun_out/timeline_cfg3/timeline.txt
