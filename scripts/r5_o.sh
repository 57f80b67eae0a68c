#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
python bench.py --steps 6 --warmup 2 --other-configs 0 --second-genome 0 --cpu-pairs 0 --vcf-reduce 0 --file-steps 0 --pcie-steps 18 > gpurun_out/r5_o.json 2> gpurun_out/r5_o.err
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_o.json') if x.startswith('{')][-1]
d=json.loads(l); p=d.get('value_pcie_inclusive',{})
print(d['ms_per_step'], 'pcie', p.get('ms_per_step'), 'native', {k: v for k, v in (p.get('system_runtime') or {}).items() if k != 'note'})
P
tail -3 gpurun_out/r5_o.err | cut -c1-300
