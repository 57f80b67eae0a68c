#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
SECONDS=0
timeout 3000 python -m pytest tests -m gpu -q --timeout 2800 -p no:cacheprovider -x -k "large_batch or full_size or packed_host or overlapped or bench_launches or sharded_run or native_cli_several or rccl_one_rank" > gpurun_out/r5_i_test.log 2>&1
echo "tests: $SECONDS s" >> gpurun_out/r5_i_test.log; tail -5 gpurun_out/r5_i_test.log | cut -c1-300
TAG=r5i bash scripts/timeline.sh > /dev/null 2>&1; head -14 gpurun_out/timeline_r5i/timeline.txt; grep -E "k_finish|k_reduce_stats|k_avg_walk" gpurun_out/timeline_r5i/timeline.txt
bash scripts/timeline_pcie.sh > gpurun_out/r5_i_pcie.log 2>&1; head -3 gpurun_out/r5_i_pcie.log
python bench.py --steps 10 --warmup 2 --other-configs 0 --second-genome 0 --cpu-pairs 0 --file-steps 0 --vcf-reduce 0 > gpurun_out/r5_bench7.json 2> gpurun_out/r5_bench7.err
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_bench7.json') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])
print(d.get('value_pcie_inclusive',{}).get('value'), d.get('value_pcie_inclusive',{}).get('ms_per_step'))
P
