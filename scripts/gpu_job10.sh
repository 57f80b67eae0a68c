#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
SECONDS=0; python bench.py > gpurun_out/r2_bench_full.json 2> gpurun_out/r2_bench_full.err; echo "Elapsed $SECONDS s"

python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r2_bench_full.json") if l.startswith("{")][-1])
print(json.dumps({k:d[k] for k in d if k not in ("roofline","config","cpu_baseline","other_genome","vcf_reduce")})[:900])
r=d["roofline"]; print("ROOF", {k:r[k] for k in ("kernel","achieved","frac","traffic","avg_launch_ms","basis","algorithmic_frac")})
print("REQ", r["request_rate"]); print("PK", r["per_kernel"])
print("VCF", d.get("vcf_reduce")); print("CPU", d.get("cpu_baseline")); print("OTHER", json.dumps(d.get("other_genome"))[:1500])
P
