#!/bin/bash
# round 5: the compact planes — every test that touches the profile, then the bench's -vcf leg with stage times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
SECONDS=0
timeout 3000 python -m pytest tests -m gpu -q --timeout 2800 -p no:cacheprovider -x -k "profile or vcf or config4 or shards or sharded or run_module or bench_launches or three_shards or with_the_profile" 2>&1 | tail -15 > gpurun_out/r5_c_test.log
echo "tests: $SECONDS s" >> gpurun_out/r5_c_test.log; tail -4 gpurun_out/r5_c_test.log
SECONDS=0; MCX_TIMING=1 python bench.py --steps 6 --warmup 2 --other-configs 0 --second-genome 0 --cpu-pairs 0 --file-steps 0 --pcie-steps 0 > gpurun_out/r5_bench2.json 2> gpurun_out/r5_bench2.err; echo "bench: $SECONDS s"
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_bench2.json') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])
print({k:v for k,v in d['vcf_reduce'].items()})
P
grep -E "mcx profile|mcx_ctx_create|profile\]" gpurun_out/r5_bench2.err | tail -12
tail -3 gpurun_out/r5_bench2.err
