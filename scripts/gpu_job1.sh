#!/bin/bash
# round-2 check: N>1 bench path on one GPU (gloo, shared device), then the default bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
MCX_BENCH_SHARE_GPU=1 MCX_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --genome-mbp 200 --batch-pairs 500000 --steps 2 --warmup 1 --cpu-pairs 0 --second-genome 0 --pcie-steps 2 > gpurun_out/r2_bench_2rank.json 2> gpurun_out/r2_bench_2rank.err
tail -c 600 gpurun_out/r2_bench_2rank.err
MCX_TIMING=1 timeout 1500 python bench.py --steps 5 --warmup 1 > gpurun_out/r2_bench_a.json 2> gpurun_out/r2_bench_a.err
tail -c 1500 gpurun_out/r2_bench_a.err
