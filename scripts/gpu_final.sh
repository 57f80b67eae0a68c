#!/bin/bash
# end-of-round evidence: the GPU suite, the rocprofv3 summaries of both workloads, the default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 1700 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r2_pytest_final.log; tail -4 gpurun_out/r2_pytest_final.log
bash scripts/collect_profile.sh r2b human > gpurun_out/r2b.log 2>&1
bash scripts/collect_profile.sh r2b_uniform uniform > gpurun_out/r2b_uniform.log 2>&1
mkdir -p profiles/round2
cp gpurun_out/r2b/summary.json profiles/round2/summary_human.json; cp gpurun_out/r2b_uniform/summary.json profiles/round2/summary_uniform.json
SECONDS=0; python bench.py --steps 10 --warmup 2 > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; echo "bench: $SECONDS s"
tail -c 600 gpurun_out/r2_bench_final.json
