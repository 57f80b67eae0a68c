#!/bin/bash
# round 6, the last GPU minutes: more fuzz rounds with new seeds on the final build — with -vcf (mate rescue, the profile, the caller), wide reads, single-end heavy
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
run() { timeout 900 python3 scripts/fuzz_parity.py "$@" > gpurun_out/fuzz_one.txt 2>&1; grep -v ": ok" gpurun_out/fuzz_one.txt | cut -c1-400; }
( echo "== default (with -vcf), seed 80801"; run --rounds 70 --seed 80801
  echo "== --wide (with -vcf), seed 80802"; run --rounds 40 --seed 80802 --wide
  echo "== large-batch paths, --wide, with -vcf, seed 80803"; MCX_ORDER_MIN=1 MCX_DP_LANE_ALWAYS=1 run --rounds 30 --seed 80803 --wide ) > gpurun_out/fuzz_r6_final2.txt 2>&1
cat gpurun_out/fuzz_r6_final2.txt
