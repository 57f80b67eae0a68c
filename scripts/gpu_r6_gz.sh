#!/bin/bash
# the .gz rate at a size that shows it: 4 M pairs (two files of 1.3 GB of text each)
mkdir -p gpurun_out/r6i; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
timeout 1500 python3 scripts/gz_rate.py --pairs 4000000 > gpurun_out/r6i/gz_rate_4m.json 2> gpurun_out/r6i/gz_rate_4m.err < /dev/null
cat gpurun_out/r6i/gz_rate_4m.json; tail -3 gpurun_out/r6i/gz_rate_4m.err
