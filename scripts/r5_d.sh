#!/bin/bash
# round 5: kernel trace of the -vcf leg (which kernels make up the bookkeeping's time), and the two tests the last run did not reach
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
out=gpurun_out/r5_vcf_trace; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --steps 2 --warmup 1 --cpu-pairs 0 --vcf-reduce 1 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0 > $out/bench.json 2> $out/bench.err < /dev/null
f=$(find $out -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r5_vcf_kernel_stats.csv; head -45 "$f" | cut -c1-200
find $out -name "*kernel_trace.csv" -size +1k | head -2
rm -rf $out/*kernel_trace.csv $out/*/*kernel_trace.csv 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q --timeout 1400 -p no:cacheprovider -x -k "bench_launches or rccl_one_rank or run_module or native_cli_several" 2>&1 | tail -5
