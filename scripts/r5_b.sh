#!/bin/bash
# round 5: the straight-line path under -vcf — parity tests that touch it, then the bench's -vcf leg
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
SECONDS=0
timeout 3000 python -m pytest tests -m gpu -q --timeout 2800 -p no:cacheprovider -x -k "alignment_profile or vcf_equals or large_batch or with_the_profile or config4 or cli_sam_and_vcf or degenerate or profile_runs or sam_equals_reference" 2>&1 | tail -15 > gpurun_out/r5_b_test.log
echo "tests: $SECONDS s" >> gpurun_out/r5_b_test.log; tail -4 gpurun_out/r5_b_test.log
SECONDS=0; python bench.py --steps 10 --warmup 2 --other-configs 0 --second-genome 0 --cpu-pairs 0 --file-steps 0 --pcie-steps 0 > gpurun_out/r5_bench1.json 2> gpurun_out/r5_bench1.err; echo "bench: $SECONDS s"
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_bench1.json') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])
print({k:v for k,v in d['vcf_reduce'].items() if k!='call_variants'})
P
tail -5 gpurun_out/r5_bench1.err
