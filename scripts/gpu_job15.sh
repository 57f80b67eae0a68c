#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/j15; ulimit -c 0
for v in prio noprio prio noprio; do
if [ $v = noprio ]; then export MCX_TIER1_NO_PRIORITY=1; else unset MCX_TIER1_NO_PRIORITY; fi
timeout 600 python3 bench.py --steps 6 --warmup 2 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 > gpurun_out/j15/$v.log 2>&1 < /dev/null
echo "$v rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/j15/$v.log | head -1; grep -o '"stage_ms_per_step": {[^}]*}' gpurun_out/j15/$v.log | head -1
done
