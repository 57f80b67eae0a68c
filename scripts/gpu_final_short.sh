#!/bin/bash
# the headline workload's evidence again after a change that leaves the other workloads' kernels alone: its rocprofv3 summary, the
# default bench line, the GPU suite (scripts/gpu_final.sh collects everything)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out profiles/round4; ulimit -c 0
R=profiles/round4
bash scripts/collect_profile.sh r4_human human > gpurun_out/r4_human.log 2>&1
cp gpurun_out/r4_human/summary.json $R/summary_human.json; cp gpurun_out/r4_human/kernel_stats.csv $R/rocprofv3_kernel_stats_human.csv
SECONDS=0; python bench.py --steps 10 --warmup 2 > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err; echo "bench: $SECONDS s"
cp gpurun_out/r4_bench_final.json $R/bench_final.json
python - <<'P'
import json
d = json.load(open("gpurun_out/r4_bench_final.json"))
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["avg_launch_ms"], r["request_rate"]["k_seed"], d["config"]["index_hbm_gb"])
P
timeout 3000 python -m pytest tests -m gpu -q --timeout 2400 -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r4_pytest_final.log; tail -4 gpurun_out/r4_pytest_final.log
mkdir -p gpurun_out/r4_profiles; cp $R/summary_human.json $R/rocprofv3_kernel_stats_human.csv $R/bench_final.json gpurun_out/r4_profiles/
