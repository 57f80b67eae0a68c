#!/bin/bash
# end-of-round evidence (round 4): the rocprofv3 summaries of the four workloads (BASELINE config 3 on both synthetic genomes, configs 5 and 2),
# the default bench line, a full bench batch against the compiled reference, run-to-run determinism, the GPU suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out profiles/round4; ulimit -c 0
R=profiles/round4
bash scripts/collect_profile.sh r4_human human > gpurun_out/r4_human.log 2>&1
bash scripts/collect_profile.sh r4_uniform uniform > gpurun_out/r4_uniform.log 2>&1
bash scripts/profile_configs.sh r4 > gpurun_out/r4_configs.log 2>&1
cp gpurun_out/r4_human/summary.json $R/summary_human.json; cp gpurun_out/r4_uniform/summary.json $R/summary_uniform.json
cp gpurun_out/r4_cfg5/summary.json $R/summary_cfg5.json; cp gpurun_out/r4_cfg2/summary.json $R/summary_cfg2.json
cp gpurun_out/r4_human/kernel_stats.csv $R/rocprofv3_kernel_stats_human.csv; cp gpurun_out/r4_uniform/kernel_stats.csv $R/rocprofv3_kernel_stats_uniform.csv
cp gpurun_out/r4_cfg5/kernel_stats.csv $R/rocprofv3_kernel_stats_cfg5.csv; cp gpurun_out/r4_cfg2/kernel_stats.csv $R/rocprofv3_kernel_stats_cfg2.csv
SECONDS=0; python bench.py --steps 10 --warmup 2 > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err; echo "bench: $SECONDS s"
cp gpurun_out/r4_bench_final.json $R/bench_final.json
tail -c 800 gpurun_out/r4_bench_final.json
timeout 1800 python scripts/full_batch_parity.py --out $R/full_batch_parity.json > gpurun_out/r4_full_batch_parity.log 2>&1; tail -c 300 gpurun_out/r4_full_batch_parity.log
cp $R/full_batch_parity.json gpurun_out/r4_full_batch_parity.json
timeout 900 python scripts/determinism.py > gpurun_out/r4_determinism.json 2> gpurun_out/r4_determinism.err; tail -c 400 gpurun_out/r4_determinism.json; cp gpurun_out/r4_determinism.json $R/determinism_mapping.json
timeout 3000 python -m pytest tests -m gpu -q --timeout 2400 -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r4_pytest_final.log; tail -4 gpurun_out/r4_pytest_final.log
mkdir -p gpurun_out/r4_profiles; cp $R/*.json $R/*.csv gpurun_out/r4_profiles/ 2>/dev/null
