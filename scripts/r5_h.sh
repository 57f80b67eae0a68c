#!/bin/bash
# round 5: the whole GPU suite on the current build, then the default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
SECONDS=0
timeout 3300 python -m pytest tests -m gpu -q --timeout 2800 -p no:cacheprovider > gpurun_out/r5_h_test.log 2>&1
echo "tests: $SECONDS s" >> gpurun_out/r5_h_test.log; tail -8 gpurun_out/r5_h_test.log | cut -c1-300
SECONDS=0; MCX_TIMING=1 python bench.py > gpurun_out/r5_bench6.json 2> gpurun_out/r5_bench6.err; echo "bench: $SECONDS s"
grep "no room in HBM" gpurun_out/r5_bench6.err | head -3
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_bench6.json') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])
print({k:v for k,v in d['vcf_reduce'].items() if k not in ('call_variants','note','reduce')})
print(d.get('value_pcie_inclusive',{}).get('value'), d.get('value_file_to_file',{}).get('value'), d.get('cpu_baseline'))
for o in d.get('other_configs',[]): print(o.get('config'), o.get('value'), o.get('ms_per_step'), o.get('roofline',{}).get('launch_bound'), o.get('error'))
P
