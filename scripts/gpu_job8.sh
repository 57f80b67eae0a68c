#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
MCX_TIMING=1 timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 1500 -p no:cacheprovider -k "config5" 2>&1 | grep -v "^\[run_pairs\]\|^\[tier" | tail -40 > gpurun_out/r2_pytest11.log; tail -30 gpurun_out/r2_pytest11.log | cut -c1-600
