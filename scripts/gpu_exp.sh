#!/bin/bash
# experiments on a small human-like genome (sourced by the experiment lists): run <name> <lib variant> ENV=VAL ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
run() {
  name=$1; lib=$2; shift 2
  cp mapcaller_amd/libmcx_$lib.so mapcaller_amd/libmcx.so
  env "$@" MCX_TIMING=1 timeout 600 python bench.py --genome-mbp 300 --batch-pairs 1000000 --steps 3 --warmup 1 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 ${GENOME:+--genome $GENOME} > gpurun_out/exp_$name.json 2> gpurun_out/exp_$name.err
  python - <<P
import json
d=json.loads([l for l in open("gpurun_out/exp_$name.json") if l.startswith("{")][-1])
print("$name", round(d["value"]/1e6,1), d["ms_per_step"], d["stage_ms_per_step"], "tier1", d["tier1_pairs"])
P
  grep -E "tier 1|run_fast" gpurun_out/exp_$name.err | tail -2
}
