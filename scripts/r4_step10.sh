#!/bin/bash
# the straight-line path with its DP problems solved one per lane (collect / solve / replay): parity, then the step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 1400 -p no:cacheprovider -x -k "large_batch_paths or large_batch_machinery or keeps_its_shape or full_size or config2 or fresh_seeded" 2>&1 | tail -15 > gpurun_out/r4_pytest_simple2.log
tail -8 gpurun_out/r4_pytest_simple2.log
MCX_TIMING=1 timeout 900 python scripts/ab_env.py --steps 4 --rounds 2 "" "MCX_SIMPLE_NO_DP=1" > gpurun_out/r4_ab8.txt 2> gpurun_out/r4_ab8.err
cat gpurun_out/r4_ab8.txt; grep "run_pairs\] pairs 4000000" gpurun_out/r4_ab8.err | head -2
timeout 600 python scripts/ab_env.py --steps 4 "MCX_SEED_FM_BUDGET=3" "MCX_SEED_FM_BUDGET=4" "MCX_SEED_FM_BUDGET=8" "MCX_SEED_FM_BUDGET=12" > gpurun_out/r4_ab8_budget.txt 2> gpurun_out/r4_ab8_budget.err
cut -c1-330 gpurun_out/r4_ab8_budget.txt
