#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "test_sam_equals_reference or fused_kernel or alignment_profile or vcf_equals or fresh_seeded or degenerate or ragged or long_cigars or maximum_read or overlapped" 2>&1 | tail -30 > gpurun_out/r2_pytest5.log; tail -8 gpurun_out/r2_pytest5.log | cut -c1-600
MCX_TIMING=1 timeout 900 python bench.py --steps 3 --warmup 1 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 6 > gpurun_out/r2_bench_d.json 2> gpurun_out/r2_bench_d.err
grep -E "tier 1|run_fast" gpurun_out/r2_bench_d.err | tail -4
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r2_bench_d.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["stage_ms_per_step"], d["tier1_pairs"]); print(d.get("value_pcie_inclusive"))
o=d.get("other_genome"); print(o and (o["value"], o["ms_per_step"], o["stage_ms_per_step"]))
P
