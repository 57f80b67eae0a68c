#!/bin/bash
# long fuzz runs beyond the suite's: wide rounds (reads up to 900 bp, more switches), every other round over the pair records;
# then the same generator with what only a large batch switches on forced onto the small ones (MCX_ORDER_MIN=1: the straight-line
# path with its DP problems — open to the pairs when no alignment profile is kept: --no-vcf —, the order lists; MCX_DP_LANE_ALWAYS=1: the lane DP kernels)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
if [ "$1" != "large-batch-paths" ]; then
( timeout 1500 python scripts/fuzz_parity.py --rounds 100 --seed 4242 --wide 2>&1 | tail -3
  timeout 900 python scripts/fuzz_parity.py --rounds 40 --seed 777 --cli-args "-devices 0,0,0 -batch 400" 2>&1 | tail -2
  timeout 600 python scripts/fuzz_parity.py --rounds 40 --seed 31337 --cli-args "-two_base -batch 1000" 2>&1 | tail -2 ) > gpurun_out/fuzz_long_runs.txt 2>&1
cat gpurun_out/fuzz_long_runs.txt
fi
export MCX_ORDER_MIN=1 MCX_DP_LANE_ALWAYS=1
( timeout 1500 python scripts/fuzz_parity.py --rounds 150 --seed 9001 --no-vcf 2>&1 | tail -2
  timeout 1500 python scripts/fuzz_parity.py --rounds 100 --seed 9002 --no-vcf --wide 2>&1 | tail -2
  timeout 900 python scripts/fuzz_parity.py --rounds 40 --seed 9003 2>&1 | tail -2 ) > gpurun_out/fuzz_large_batch_paths.txt 2>&1
cat gpurun_out/fuzz_large_batch_paths.txt
