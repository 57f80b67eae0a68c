#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
for b in 48 0 16 128; do
  MCX_D2H_BLOCKS=$b python bench.py --steps 6 --warmup 2 --other-configs 0 --second-genome 0 --cpu-pairs 0 --file-steps 0 --vcf-reduce 0 --pcie-steps 10 > gpurun_out/r5_j_$b.json 2> gpurun_out/r5_j_$b.err
  python - $b <<'P'
import json,sys
l=[x for x in open('gpurun_out/r5_j_%s.json'%sys.argv[1]) if x.startswith('{')][-1]
d=json.loads(l)
print('d2h_blocks', sys.argv[1], 'step', d['ms_per_step'], 'cluster', d['stage_ms_per_step']['cluster'], 'pcie', d.get('value_pcie_inclusive',{}).get('ms_per_step'), d.get('value_pcie_inclusive',{}).get('value'))
P
done
MCX_KTAB_K=16 python bench.py --steps 6 --warmup 2 --other-configs 0 --second-genome 0 --cpu-pairs 0 --file-steps 0 --vcf-reduce 0 --pcie-steps 0 > gpurun_out/r5_j_k16.json 2> gpurun_out/r5_j_k16.err
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_j_k16.json') if x.startswith('{')][-1]
d=json.loads(l)
print('ktab 16:', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['config']['index_hbm_gb'], d['per_read'])
P
