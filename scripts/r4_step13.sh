#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 1400 -p no:cacheprovider -x -k "large_batch_paths or large_batch_machinery or keeps_its_shape" 2>&1 | tail -5
timeout 900 python scripts/ab_env.py --steps 6 "" > gpurun_out/r4_ab10.txt 2> gpurun_out/r4_ab10.err
cut -c1-330 gpurun_out/r4_ab10.txt
TAG=r4u bash scripts/timeline.sh > /dev/null 2>&1; awk '$1 >= 4.0 && $1 <= 6.2' gpurun_out/timeline_r4u/timeline.txt | cut -c1-110
