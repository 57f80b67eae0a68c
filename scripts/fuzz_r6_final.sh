#!/bin/bash
# round 6, after its last device changes (the lane DP's shared walk windows, the one edge word per row, the rescue's seeds by all lanes, the growths): fuzz rounds with new seeds,
# the CLI against the oracle.  A round that differs leaves its files under gpurun_out/fuzz_fail and is printed here.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
run() { timeout 1200 python3 scripts/fuzz_parity.py "$@" > gpurun_out/fuzz_one.txt 2>&1; grep -v ": ok" gpurun_out/fuzz_one.txt | cut -c1-400; }
( echo "== --wide, seed 70701"; run --rounds 40 --seed 70701 --wide
  echo "== large-batch paths (every DP list on the lane kernels, order lists), --no-vcf, seed 70702"; MCX_ORDER_MIN=1 MCX_DP_LANE_ALWAYS=1 run --rounds 60 --seed 70702 --no-vcf
  echo "== large-batch paths, --no-vcf --wide, seed 70703"; MCX_ORDER_MIN=1 MCX_DP_LANE_ALWAYS=1 run --rounds 40 --seed 70703 --no-vcf --wide
  echo "== large-batch paths with -vcf, seed 70704"; MCX_ORDER_MIN=1 MCX_DP_LANE_ALWAYS=1 run --rounds 30 --seed 70704
  echo "== three shards, seed 70705"; run --rounds 20 --seed 70705 --cli-args "-devices 0,0,0 -batch 400"
  echo "== small large tier that grows (MCX_TIER1_START_GB=0.05), --wide, seed 70706"; MCX_TIER1_START_GB=0.05 run --rounds 20 --seed 70706 --wide ) > gpurun_out/fuzz_r6_final.txt 2>&1
cat gpurun_out/fuzz_r6_final.txt
