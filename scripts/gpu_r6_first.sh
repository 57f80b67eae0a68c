#!/bin/bash
# round 6, first GPU call: the default bench line (does it parse?) and the GPU suite with per-test durations
mkdir -p gpurun_out/r6a
python3 bench.py > gpurun_out/r6a/bench_stdout.txt 2> gpurun_out/r6a/bench_stderr.txt < /dev/null
tail -c 7000 gpurun_out/r6a/bench_stdout.txt
cp gpurun_out/bench_detail*.json gpurun_out/r6a/ 2>/dev/null
timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=60 > gpurun_out/r6a/gpu_suite.log 2>&1 < /dev/null
tail -80 gpurun_out/r6a/gpu_suite.log
