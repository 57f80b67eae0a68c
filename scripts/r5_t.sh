#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
echo skip tests
for v in A=1 MCX_NO_PROF_OVERLAP=1; do
env MCX_TIMING=1 $v python bench.py --steps 3 --warmup 1 --other-configs 0 --second-genome 0 --cpu-pairs 0 --vcf-reduce 1 --pcie-steps 0 --file-steps 0 > gpurun_out/r5_t.json 2> gpurun_out/r5_t.err
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_t.json') if x.startswith('{')][-1]
d=json.loads(l); v=d.get('vcf_reduce',{})
print(d['ms_per_step'], {k: v[k] for k in v if 'ms' in k or 'error' in k or 'hbm' in k or 'slice' in k})
P
grep -c "queued behind" gpurun_out/r5_t.err; grep "no room in HBM" gpurun_out/r5_t.err | head -3
done
