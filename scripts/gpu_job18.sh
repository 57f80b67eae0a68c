#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/j18; ulimit -c 0
run() { timeout 600 python3 bench.py --steps 5 --warmup 2 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 > gpurun_out/j18/$1.log 2>&1 < /dev/null
echo "$1 rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/j18/$1.log | head -1; grep -o '"stage_ms_per_step": {[^}]*}' gpurun_out/j18/$1.log | head -1; }
run b1024
MCX_RESCUE_BLOCKS=512 run b512
MCX_RESCUE_BLOCKS=2048 run b2048
MCX_BRANCH_NORMAL=1 run b1024_normal
MCX_BRANCH_NORMAL=1 MCX_RESCUE_BLOCKS=512 run b512_normal
