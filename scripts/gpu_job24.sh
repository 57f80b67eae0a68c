#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/j24; ulimit -c 0
run() { timeout 600 python3 bench.py --steps 4 --warmup 2 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 > gpurun_out/j24/$1.log 2>&1 < /dev/null
echo "$1 rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/j24/$1.log | head -1; grep -o '"stage_ms_per_step": {[^}]*}' gpurun_out/j24/$1.log | head -1; grep -o '"tier1_pairs": [0-9]*' gpurun_out/j24/$1.log | head -1; grep -i "error" gpurun_out/j24/$1.log | head -3; }
MCX_TIER0_CAPS=56,12,64,1024 run c56
MCX_TIER0_CAPS=32,8,48,512 run c32
MCX_TIER0_CAPS=16,6,24,256 run c16
