#!/bin/bash
# round 5: the batch tail — trajectory / parity tests, then the bench's timed region and the -vcf leg; the 8-rank launch with its stderr kept
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
SECONDS=0
timeout 3000 python -m pytest tests -m gpu -q --timeout 2800 -p no:cacheprovider -x -k "sam_equals_reference or trajectory or large_batch or full_size or config4 or boundary or fresh_seeded or bench_workload or one_context or config5 or config2" > gpurun_out/r5_f_test.log 2>&1
echo "tests: $SECONDS s" >> gpurun_out/r5_f_test.log; tail -4 gpurun_out/r5_f_test.log
MCX_BENCH_SHARE_GPU=1 MCX_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --genome uniform --genome-mbp 20 --contigs 4 --repeats 50 --batch-pairs 40000 --cpu-pairs 0 --vcf-reduce 1 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0 > gpurun_out/r5_f_8rank.json 2> gpurun_out/r5_f_8rank.err; echo "8 ranks rc $?"; grep -v "^\[Gloo\]\|amdgpu.ids" gpurun_out/r5_f_8rank.err | grep -B2 -A12 "Traceback\|Error\|error" | head -60
python bench.py --steps 10 --warmup 2 --other-configs 0 --second-genome 0 --cpu-pairs 0 --file-steps 0 --pcie-steps 0 > gpurun_out/r5_bench4.json 2> gpurun_out/r5_bench4.err
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_bench4.json') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])
print({k:v for k,v in d['vcf_reduce'].items() if k not in ('call_variants','note','reduce')})
print(d['roofline']['kernel'], d['roofline']['frac'], list(d['roofline']['per_kernel'].keys())[:40])
P
