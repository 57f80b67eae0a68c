#!/bin/bash
# the -vcf surface after a change to the bookkeeping: its parity tests, then the bench's -vcf leg under the kernel trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3_vcf; ulimit -c 0
[ -n "$SKIP_TESTS" ] || timeout 1200 python -m pytest tests -m gpu -q --timeout 1000 -p no:cacheprovider -k "profile or vcf or shard or variant" 2>&1 | tail -8 > gpurun_out/r3_vcf/pytest.log; tail -5 gpurun_out/r3_vcf/pytest.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_vcf/kt -o kt -- python3 bench.py --steps 1 --warmup 1 --cpu-pairs 0 --vcf-reduce 1 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0 > gpurun_out/r3_vcf/kt.log 2>&1 < /dev/null
find gpurun_out/r3_vcf/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3_vcf/kernel_stats.csv \;
find gpurun_out/r3_vcf/kt -name "*kernel_trace.csv" -exec python3 scripts/slice_kernels.py {} \; > gpurun_out/r3_vcf/slices.txt 2>&1
rm -rf gpurun_out/r3_vcf/kt
grep -o '"vcf_reduce".*' gpurun_out/r3_vcf/kt.log | cut -c1-1200
head -30 gpurun_out/r3_vcf/kernel_stats.csv | cut -c1-150
cat gpurun_out/r3_vcf/slices.txt
