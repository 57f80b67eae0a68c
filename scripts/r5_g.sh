#!/bin/bash
# round 5: why the -vcf leg's batch went from 30 to 38 ms — kernel trace of the leg, and its own stage prints
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
out=gpurun_out/r5_vcf_trace2; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --vcf-reduce 1 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0 > $out/bench.json 2> $out/bench.err < /dev/null
f=$(find $out -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r5_vcf_kernel_stats2.csv
rm -rf $out/*kernel_trace.csv $out/*/*kernel_trace.csv 2>/dev/null
MCX_TIMING=1 python bench.py --steps 3 --warmup 1 --other-configs 0 --second-genome 0 --cpu-pairs 0 --file-steps 0 --pcie-steps 0 > gpurun_out/r5_bench5.json 2> gpurun_out/r5_bench5.err
grep -E "mcx profile\] mapping" gpurun_out/r5_bench5.err | tail -8
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_bench5.json') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'])
print({k:v for k,v in d['vcf_reduce'].items() if k not in ('call_variants','note','reduce')})
P
