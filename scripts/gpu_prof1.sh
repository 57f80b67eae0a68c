#!/bin/bash
# kernel trace of the bench incl. the -vcf leg (round 2 look at where the profile time goes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/prof_vcf; ulimit -c 0
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_vcf/kt -o kt -- python3 bench.py --steps 2 --warmup 1 --cpu-pairs 0 --vcf-reduce 1 --pcie-steps 0 --second-genome 0 > gpurun_out/prof_vcf/kt.log 2>&1 < /dev/null
find gpurun_out/prof_vcf/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_vcf/kernel_stats.csv \;
rm -rf gpurun_out/prof_vcf/kt
head -45 gpurun_out/prof_vcf/kernel_stats.csv | cut -d, -f1-8 | cut -c1-200
tail -c 1500 gpurun_out/prof_vcf/kt.log | grep -o '"vcf_reduce".*' | cut -c1-900
