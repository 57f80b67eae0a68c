#!/bin/bash
# quick A/B: bench on both genomes, 2 steps each, stage times only
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
tag=${1:-q}
MCX_TIMING=1 timeout 900 python bench.py --steps 2 --warmup 1 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps ${PCIE:-0} > gpurun_out/r2_bench_$tag.json 2> gpurun_out/r2_bench_$tag.err
grep -E "tier 1|run_fast" gpurun_out/r2_bench_$tag.err | tail -3
python - <<P
import json
d=json.loads([l for l in open("gpurun_out/r2_bench_$tag.json") if l.startswith("{")][-1])
print("human  ", round(d["value"]/1e6,1), d["ms_per_step"], d["stage_ms_per_step"], d["tier1_pairs"]); print(d.get("value_pcie_inclusive"))
o=d.get("other_genome"); print("uniform", o and (round(o["value"]/1e6,1), o["ms_per_step"], o["stage_ms_per_step"]))
P
