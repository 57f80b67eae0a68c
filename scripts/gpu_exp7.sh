cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
MCX_TIER1_HIST=1 MCX_TIMING=1 timeout 900 python bench.py --steps 1 --warmup 1 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 > gpurun_out/r2_bench_n.json 2> gpurun_out/r2_bench_n.err
grep -E "tier 1" gpurun_out/r2_bench_n.err | tail -6
