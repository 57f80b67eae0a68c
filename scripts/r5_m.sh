#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
SECONDS=0
timeout 3000 python -m pytest tests -m gpu -q --timeout 2800 -p no:cacheprovider -x -k "packed_host or overlapped or input_side or cli or bgzf or one_context or file_path or sam_equals_reference or ragged or long_cigars or degenerate or maximum_read or sharded_run or native_cli" > gpurun_out/r5_m_test.log 2>&1
echo "tests: $SECONDS s" >> gpurun_out/r5_m_test.log; tail -5 gpurun_out/r5_m_test.log | cut -c1-300
MCX_TIMING=1 python bench.py --steps 10 --warmup 2 --other-configs 0 --second-genome 0 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 12 > gpurun_out/r5_bench9.json 2> gpurun_out/r5_bench9.err
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_bench9.json') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])
print('pcie', d.get('value_pcie_inclusive',{}).get('value'), d.get('value_pcie_inclusive',{}).get('ms_per_step'), d.get('value_pcie_inclusive',{}).get('d2h_bytes_per_read'))
f=d.get('value_file_to_file',{}); print('files', f.get('value'), f.get('seconds'), f.get('without_sam_output',{}).get('value'))
P
grep "mcx_map_files\] busy" gpurun_out/r5_bench9.err | tail -3
