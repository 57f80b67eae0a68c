#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/j30; ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -x -q -k "large_batch or full_size or fuzz_rounds_equal" 2>&1 | tail -4
run() { timeout 600 python3 bench.py --steps 5 --warmup 2 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 > gpurun_out/j30/$1.log 2>&1 < /dev/null
echo "$1 rc=$?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/j30/$1.log | head -1; grep -o '"stage_ms_per_step": {[^}]*}' gpurun_out/j30/$1.log | head -1; grep -i "error" gpurun_out/j30/$1.log | head -3; }
run base
for w in 6 8; do MCX_LIB=$GRAFT_REPO_ROOT/mapcaller_amd/libmcx_r$w.so run r$w; done
