#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/j31; ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -x -q -k "vcf or profile or variant or cli" 2>&1 | tail -4
MCX_TIMING=1 timeout 900 python3 bench.py --steps 1 --warmup 1 --cpu-pairs 0 --vcf-reduce 1 --pcie-steps 0 --second-genome 0 > gpurun_out/j31/v.log 2>&1 < /dev/null
grep "mcx_call_variants\]" gpurun_out/j31/v.log | tail -8; grep -o '"call_variants": {[^}]*}' gpurun_out/j31/v.log
