#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/prof_vcf2; ulimit -c 0
MCX_TIMING=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_vcf2/kt -o kt -- python3 bench.py --steps 1 --warmup 1 --cpu-pairs 0 --vcf-reduce 1 --pcie-steps 0 --second-genome 0 > gpurun_out/prof_vcf2/kt.log 2>&1 < /dev/null
find gpurun_out/prof_vcf2/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_vcf2/kernel_stats.csv \;
rm -rf gpurun_out/prof_vcf2/kt
grep "\[profile\]" gpurun_out/prof_vcf2/kt.log | tail -5
grep -o '"vcf_reduce".*' gpurun_out/prof_vcf2/kt.log | cut -c1-500
python3 - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_vcf2/kernel_stats.csv')))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("k_prof","k_finish","k_build(","k_cluster","radix","copyBuffer","fillBuffer")):
        print(n[:60].ljust(60), r["Calls"].rjust(5), "%9.2f ms total %8.3f avg" % (int(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
P
