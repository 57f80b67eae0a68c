#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
for v in 0 1 2; do
  MCX_EXP_D2H=$v python bench.py --steps 4 --warmup 2 --other-configs 0 --second-genome 0 --cpu-pairs 0 --file-steps 0 --vcf-reduce 0 --pcie-steps 12 2> /dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('exp_d2h', $v, 'step', d['ms_per_step'], 'pcie', d['value_pcie_inclusive']['ms_per_step'], d['value_pcie_inclusive']['value'])"
done
