#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
SECONDS=0
timeout 3000 python -m pytest tests -m gpu -q --timeout 2800 -p no:cacheprovider -x -k "packed_host or overlapped or input_side or cli or bgzf or large_batch or one_context or file_path" > gpurun_out/r5_k_test.log 2>&1
echo "tests: $SECONDS s" >> gpurun_out/r5_k_test.log; tail -5 gpurun_out/r5_k_test.log | cut -c1-300
python bench.py --steps 10 --warmup 2 --other-configs 0 --second-genome 0 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 12 > gpurun_out/r5_bench8.json 2> gpurun_out/r5_bench8.err
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_bench8.json') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])
print('pcie', d.get('value_pcie_inclusive',{}).get('value'), d.get('value_pcie_inclusive',{}).get('ms_per_step'))
f=d.get('value_file_to_file',{}); print('files', f.get('value'), f.get('seconds'), f.get('without_sam_output'))
P
TAG=r5k bash scripts/timeline.sh > /dev/null 2>&1; head -12 gpurun_out/timeline_r5k/timeline.txt; grep -E "k_finish|k_reduce_stats|k_avg_walk" gpurun_out/timeline_r5k/timeline.txt
