#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/j17; ulimit -c 0
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or fuzz or config or human" 2>&1 | tail -4
for v in split nosplit; do
if [ $v = nosplit ]; then export MCX_NO_RESCUE_BRANCH=1; else unset MCX_NO_RESCUE_BRANCH; fi
timeout 600 python3 bench.py --steps 6 --warmup 2 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 > gpurun_out/j17/$v.log 2>&1 < /dev/null
echo "$v rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/j17/$v.log | head -1; grep -o '"stage_ms_per_step": {[^}]*}' gpurun_out/j17/$v.log | head -1
done
unset MCX_NO_RESCUE_BRANCH
MCX_TIMING=1 timeout 600 python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 > gpurun_out/j17/t.log 2>&1 < /dev/null
grep "after clustering\|beside tier 0\|tier 1\] [0-9]* pairs:" gpurun_out/j17/t.log | tail -6
