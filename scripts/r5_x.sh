#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "sam_equals_reference or ragged or degenerate or maximum_read or packed_host or long_cigars or empty or odd_letters or profile_runs_equal" 2>&1 | tail -3 | cut -c1-300
for v in A=1 MCX_PACK_DIRECT=1; do
env $v python bench.py --steps 10 --warmup 2 --other-configs 0 --second-genome 0 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --file-steps 0 > gpurun_out/r5_x.json 2> gpurun_out/r5_x.err
python - $v <<'P'
import json, sys
l=[x for x in open('gpurun_out/r5_x.json') if x.startswith('{')][-1]
d=json.loads(l); print(sys.argv[1], d['value'], d['ms_per_step'], d['stage_ms_per_step']['encode'])
P
done
