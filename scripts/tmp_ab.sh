#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/j37; ulimit -c 0
run() { timeout 600 python3 bench.py --steps 6 --warmup 2 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 > gpurun_out/j37/$1.log 2>&1 < /dev/null
echo "$1 rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/j37/$1.log | head -1; grep -o '"stage_ms_per_step": {[^}]*}' gpurun_out/j37/$1.log | head -1; grep -i "error" gpurun_out/j37/$1.log | head -3; }
run a
run b
