#!/bin/bash
# round 5's fuzz runs beyond the suite's (new seeds): the CLI — a host on the system's HIP runtime, the -vcf bookkeeping queued behind the batches, 32-byte records over the
# boundary — against the oracle: wide rounds, three shards on one device, the pair records with small batches, and the large-batch paths forced onto small batches
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
( timeout 1200 python scripts/fuzz_parity.py --rounds 80 --seed 50501 --wide 2>&1 | tail -2
  timeout 700 python scripts/fuzz_parity.py --rounds 30 --seed 50502 --cli-args "-devices 0,0,0 -batch 400" 2>&1 | tail -2
  timeout 600 python scripts/fuzz_parity.py --rounds 30 --seed 50503 --cli-args "-two_base -batch 1000" 2>&1 | tail -2
  MCX_NO_PROF_OVERLAP=1 timeout 600 python scripts/fuzz_parity.py --rounds 30 --seed 50504 --cli-args "-batch 600" 2>&1 | tail -2
  MCX_ORDER_MIN=1 MCX_DP_LANE_ALWAYS=1 timeout 900 python scripts/fuzz_parity.py --rounds 60 --seed 50505 2>&1 | tail -2 ) > gpurun_out/fuzz_r5.txt 2>&1
cat gpurun_out/fuzz_r5.txt
