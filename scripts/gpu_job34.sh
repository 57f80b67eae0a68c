#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/j34; ulimit -c 0
run() { timeout 600 python3 bench.py --steps 8 --warmup 2 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 > gpurun_out/j34/$1.log 2>&1 < /dev/null
echo "$1 rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/j34/$1.log | head -1; grep -o '"tier1_pairs": [0-9]*' gpurun_out/j34/$1.log | head -1; grep -i "error" gpurun_out/j34/$1.log | head -3; }
run a
run b
MCX_NO_LATE_OVERLAP=1 run nolate
