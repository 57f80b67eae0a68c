#!/bin/bash
# round 5: sorted accumulation — profile tests, then the -vcf leg A/B (sorted / MCX_PROF_UNSORTED)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
SECONDS=0
timeout 3000 python -m pytest tests -m gpu -q --timeout 2800 -p no:cacheprovider -x -k "profile or vcf or config4 or bench_launches or rccl_one_rank or run_module or native_cli_several or sharded_run or three_shards" > gpurun_out/r5_e_test.log 2>&1
echo "tests: $SECONDS s" >> gpurun_out/r5_e_test.log; tail -6 gpurun_out/r5_e_test.log
for v in sorted unsorted; do
  if [ $v = unsorted ]; then export MCX_PROF_UNSORTED=1; fi
  python bench.py --steps 4 --warmup 2 --other-configs 0 --second-genome 0 --cpu-pairs 0 --file-steps 0 --pcie-steps 0 > gpurun_out/r5_bench3_$v.json 2> gpurun_out/r5_bench3_$v.err
  python - $v <<'P'
import json,sys
l=[x for x in open('gpurun_out/r5_bench3_%s.json'%sys.argv[1]) if x.startswith('{')][-1]
d=json.loads(l)
print(sys.argv[1], d['value'], d['ms_per_step'], d['stage_ms_per_step'])
print({k:v for k,v in d['vcf_reduce'].items() if k not in ('call_variants','note','reduce')})
P
done
