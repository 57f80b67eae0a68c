#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
python bench.py --steps 10 --warmup 2 --other-configs 0 --second-genome 0 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 12 --file-steps 0 --native-boundary 0 --batch-pairs 1000000 > gpurun_out/r5_q.json 2> gpurun_out/r5_q.err
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_q.json') if x.startswith('{')][-1]
d=json.loads(l); p=d.get('value_pcie_inclusive',{})
print(d['ms_per_step'], d['stage_ms_per_step'], 'pcie', p.get('ms_per_step'))
P
