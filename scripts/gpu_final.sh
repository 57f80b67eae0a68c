#!/bin/bash
# end-of-round evidence: the GPU suite, the rocprofv3 summaries of both workloads, the default bench line, the -vcf leg's kernels
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out profiles/round3; ulimit -c 0
timeout 1700 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r3_pytest_final.log; tail -4 gpurun_out/r3_pytest_final.log
bash scripts/collect_profile.sh r3b human > gpurun_out/r3b.log 2>&1
bash scripts/collect_profile.sh r3b_uniform uniform > gpurun_out/r3b_uniform.log 2>&1
cp gpurun_out/r3b/summary.json profiles/round3/summary_human.json; cp gpurun_out/r3b_uniform/summary.json profiles/round3/summary_uniform.json
SECONDS=0; python bench.py --steps 10 --warmup 2 > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err; echo "bench: $SECONDS s"
tail -c 600 gpurun_out/r3_bench_final.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3b_vcf
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3b_vcf/kt -o kt -- python3 bench.py --steps 1 --warmup 1 --cpu-pairs 0 --vcf-reduce 1 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0 > gpurun_out/r3b_vcf/kt.log 2>&1 < /dev/null
find gpurun_out/r3b_vcf/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3b_vcf/kernel_stats.csv \;
rm -rf gpurun_out/r3b_vcf/kt
grep -o '"vcf_reduce".*' gpurun_out/r3b_vcf/kt.log | cut -c1-400
