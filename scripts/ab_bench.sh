#!/bin/bash
# A/B of prebuilt library variants on one GPU box: scripts/ab_bench.sh base A B ...
# (each name is mapcaller_amd/libmcx_<name>.so, built beforehand; the last one stays installed as libmcx.so)
mkdir -p gpurun_out
for v in "$@"; do
  cp mapcaller_amd/libmcx_$v.so mapcaller_amd/libmcx.so
  python bench.py --steps 4 --warmup 1 --cpu-pairs 0 --vcf-reduce 0 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || echo "fail $v"
  python - <<P
import json
d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1])
print('$v', round(d['value'] / 1e6, 1), d['stage_ms_per_step'])
P
done
