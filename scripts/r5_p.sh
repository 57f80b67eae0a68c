#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "cli or input_side or bgzf or file_path or sharded_run or native_cli or io" 2>&1 | tail -3 | cut -c1-300
for v in A=1 MCX_SAM_NO_FALLOCATE=1; do
env MCX_TIMING=1 $v python bench.py --steps 3 --warmup 1 --other-configs 0 --second-genome 0 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 "$@" > gpurun_out/r5_p.json 2> gpurun_out/r5_p.err
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_p.json') if x.startswith('{')][-1]
d=json.loads(l); f=d.get('value_file_to_file',{})
print('files', f.get('value'), f.get('seconds'), f.get('first_run_seconds'), 'no sam', f.get('without_sam_output',{}).get('value'), f.get('without_sam_output',{}).get('seconds'), f.get('sam_bytes'))
P
grep "busy seconds\|wall seconds" gpurun_out/r5_p.err | tail -6 | cut -c1-260
done
