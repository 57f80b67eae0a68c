#!/bin/bash
# the CLI without -vcf against the oracle's SAM: plain rounds (the same sequence four times over: timing is the variable), wide, three shards, small batches;
# every round that is not identical is printed
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
run() { timeout 900 python scripts/fuzz_parity.py "$@" > gpurun_out/fuzz_one.txt 2>&1; grep -v ": ok" gpurun_out/fuzz_one.txt | cut -c1-400; }
( for k in 1 2 3 4; do run --rounds 60 --seed 50601 --no-vcf; done
  run --rounds 40 --seed 50602 --no-vcf --wide
  run --rounds 30 --seed 50603 --no-vcf --cli-args "-devices 0,0,0 -batch 400" ) > gpurun_out/fuzz_r5b.txt 2>&1
cat gpurun_out/fuzz_r5b.txt
