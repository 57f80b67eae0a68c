#!/bin/bash
# round 6's fuzz runs beyond the suite's (new seeds).  Every round that differs leaves its files under gpurun_out/fuzz_fail (fuzz_parity.py --keep, on by default since
# this round) and is printed here.  The CLI against the oracle: the sequence that round 5's odd round came from with the pre-pack on and off, wide rounds, three shards
# on one device, and the large-batch paths — the straight-line path, the order lists, every DP list on the two-problems-per-lane kernels — forced onto small batches.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
run() { timeout 1200 python3 scripts/fuzz_parity.py "$@" > gpurun_out/fuzz_one.txt 2>&1; grep -v ": ok" gpurun_out/fuzz_one.txt | cut -c1-400; }
( echo "== MCX_PREPACK=1, --no-vcf, seed 50601 (round 5's sequence) x2"; for k in 1 2; do MCX_PREPACK=1 run --rounds 60 --seed 50601 --no-vcf; done
  echo "== MCX_PREPACK=1, --no-vcf, seed 60601"; MCX_PREPACK=1 run --rounds 60 --seed 60601 --no-vcf
  echo "== default, --no-vcf, seed 60601"; run --rounds 60 --seed 60601 --no-vcf
  echo "== --wide, seed 60602"; run --rounds 40 --seed 60602 --wide
  echo "== three shards, seed 60603"; run --rounds 30 --seed 60603 --cli-args "-devices 0,0,0 -batch 400"
  echo "== large-batch paths, --no-vcf, seed 60604"; MCX_ORDER_MIN=1 MCX_DP_LANE_ALWAYS=1 run --rounds 60 --seed 60604 --no-vcf
  echo "== large-batch paths, --no-vcf --wide, seed 60605"; MCX_ORDER_MIN=1 MCX_DP_LANE_ALWAYS=1 run --rounds 40 --seed 60605 --no-vcf --wide
  echo "== large-batch paths with -vcf, seed 60606"; MCX_ORDER_MIN=1 MCX_DP_LANE_ALWAYS=1 run --rounds 30 --seed 60606
  echo "== large-batch paths, one DP problem per lane (MCX_DP_X1=1), seed 60604"; MCX_DP_X1=1 MCX_ORDER_MIN=1 MCX_DP_LANE_ALWAYS=1 run --rounds 30 --seed 60604 --no-vcf ) > gpurun_out/fuzz_r6.txt 2>&1
cat gpurun_out/fuzz_r6.txt
