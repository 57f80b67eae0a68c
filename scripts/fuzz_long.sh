#!/bin/bash
# long fuzz runs beyond the suite's: wide rounds (reads up to 900 bp, more switches), every other round over the pair records
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
( timeout 1500 python scripts/fuzz_parity.py --rounds 100 --seed 4242 --wide 2>&1 | tail -3
  timeout 900 python scripts/fuzz_parity.py --rounds 40 --seed 777 --cli-args "-devices 0,0,0 -batch 400" 2>&1 | tail -2
  timeout 600 python scripts/fuzz_parity.py --rounds 40 --seed 31337 --cli-args "-two_base -batch 1000" 2>&1 | tail -2 ) > gpurun_out/fuzz_long_runs.txt 2>&1
cat gpurun_out/fuzz_long_runs.txt
