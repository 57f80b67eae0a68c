#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/j33; ulimit -c 0
run() { timeout 600 python3 bench.py --steps 5 --warmup 2 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 > gpurun_out/j33/$1.log 2>&1 < /dev/null
echo "$1 rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/j33/$1.log | head -1; grep -o '"stage_ms_per_step": {[^}]*}' gpurun_out/j33/$1.log | head -1; grep -o '"tier1_pairs": [0-9]*' gpurun_out/j33/$1.log | head -1; grep -i "error" gpurun_out/j33/$1.log | head -3; }
run byhits
MCX_NO_EARLY_BY_HITS=1 run plain
timeout 1500 python -m pytest tests -m gpu -x -q -k "large_batch or full_size or fuzz_rounds_equal or human" 2>&1 | grep -E "passed|failed|rror" | tail -5
