#!/bin/bash
# round end: the whole GPU suite, the determinism record of the -vcf planes, the default bench line (gpurun -- bash scripts/gpu_round_end_r5.sh)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
SECONDS=0
timeout 2700 python -m pytest tests -m gpu -q --timeout 2400 -p no:cacheprovider > gpurun_out/r5_full_test.log 2>&1
echo "tests: $SECONDS s" >> gpurun_out/r5_full_test.log; grep -E "passed|failed" gpurun_out/r5_full_test.log | tail -3 | cut -c1-300
timeout 1500 python scripts/determinism_vcf.py > gpurun_out/r5_determinism_vcf.json 2> gpurun_out/r5_determinism_vcf.err; tail -c 400 gpurun_out/r5_determinism_vcf.json; echo
SECONDS=0
python bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err; echo "default bench: $SECONDS s"
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_bench_default.json') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
p=d.get('value_pcie_inclusive',{}); print('pcie', p.get('ms_per_step'), (p.get('system_runtime') or {}).get('ms_per_step'))
f=d.get('value_file_to_file',{}); print('files', f.get('value'), f.get('seconds'), f.get('without_sam_output',{}).get('value'))
c=d.get('cpu_baseline',{}); print('cpu', {k: c.get(k) for k in ('value','cores','hardware_threads','kind')}, (c.get('mapping_only') or {}).get('value'), (c.get('single_thread') or {}).get('value'))
v=d.get('vcf_reduce'); print('vcf', v)
o=d.get('other_configs'); print('other', [(x.get('name') or x.get('config'), x.get('value'), x.get('ms_per_step')) for x in (o or [])] if isinstance(o, list) else o)
g=d.get('other_genome'); print('genome2', g and (g.get('value'), g.get('ms_per_step')))
P
