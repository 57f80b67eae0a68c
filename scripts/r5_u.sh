#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
for v in MCX_PROF_BLOCKS=0 MCX_PROF_BLOCKS=512 MCX_PROF_BLOCKS=1024 MCX_PROF_BLOCKS=2048 MCX_PROF_BLOCKS=4096; do
env $v python bench.py --steps 5 --warmup 1 --other-configs 0 --second-genome 0 --cpu-pairs 0 --vcf-reduce 1 --pcie-steps 0 --file-steps 0 > gpurun_out/r5_u.json 2> gpurun_out/r5_u.err
python - $v <<'P'
import json, sys
l=[x for x in open('gpurun_out/r5_u.json') if x.startswith('{')][-1]
d=json.loads(l); v=d.get('vcf_reduce',{})
print(sys.argv[1], d['ms_per_step'], {k: v[k] for k in v if k in ('profile_batch_ms','same_batches_without_profile_ms','sequence_ms_per_batch_tail_included','last_batch_bookkeeping_tail_ms','error','batches_timed')})
P
done
