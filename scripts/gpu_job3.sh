#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
ulimit -c 0
timeout 1700 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r2_pytest3.log; tail -25 gpurun_out/r2_pytest3.log
MCX_TIMING=1 timeout 900 python bench.py --steps 3 --warmup 1 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 > gpurun_out/r2_bench_c.json 2> gpurun_out/r2_bench_c.err
grep -E "tier 1|run_fast" gpurun_out/r2_bench_c.err | tail -6
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r2_bench_c.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["stage_ms_per_step"], d["tier1_pairs"])
o=d.get("other_genome"); print(o and (o["value"], o["ms_per_step"], o["stage_ms_per_step"]))
P
