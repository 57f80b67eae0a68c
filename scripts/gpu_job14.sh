#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/j14; ulimit -c 0
for sl in 4000000 8000000; do
timeout 600 python3 bench.py --steps 1 --warmup 1 --cpu-pairs 0 --vcf-reduce 1 --pcie-steps 0 --second-genome 0 --vcf-slice-reads $sl > gpurun_out/j14/s$sl.log 2>&1 < /dev/null
echo "slice $sl rc=$?"; grep -o '"vcf_reduce".*' gpurun_out/j14/s$sl.log | cut -c1-420; grep -i "error\|memory" gpurun_out/j14/s$sl.log | head -3
done
