#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/j23; ulimit -c 0
timeout 600 python3 bench.py --steps 6 --warmup 2 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 > gpurun_out/j23/a.log 2>&1 < /dev/null
echo "rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/j23/a.log | head -1; grep -o '"stage_ms_per_step": {[^}]*}' gpurun_out/j23/a.log | head -1
MCX_TIMING=1 timeout 600 python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 > gpurun_out/j23/t.log 2>&1 < /dev/null
grep "after clustering\|beside tier 0\|tier 1\] [0-9]* pairs:" gpurun_out/j23/t.log | tail -9
