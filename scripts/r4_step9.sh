#!/bin/bash
# pair records: golden SAM + the full-size genome tests, then the step with and without them
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 1400 -p no:cacheprovider -x -k "test_sam_equals_reference or full_size or large_batch_machinery or keeps_its_shape or config5 or config2 or ecoli" 2>&1 | tail -15 > gpurun_out/r4_pytest_pairs.log
tail -8 gpurun_out/r4_pytest_pairs.log
MCX_RANK2_CHECK=2000000 MCX_TIMING=1 timeout 900 python scripts/ab_env.py --steps 4 --rounds 2 "" "MCX_SEED_ONE_BASE=1" > gpurun_out/r4_ab7.txt 2> gpurun_out/r4_ab7.err
cat gpurun_out/r4_ab7.txt; grep -i "hbm\|index" gpurun_out/r4_ab7.err | head -5
MCX_RANK2_CHECK=2000000 timeout 900 python scripts/ab_env.py --steps 2 --rlen 250 --ins 0.025 --dele 0.025 --alg nw "" "MCX_SEED_ONE_BASE=1" > gpurun_out/r4_ab7_cfg5.txt 2> gpurun_out/r4_ab7_cfg5.err
cat gpurun_out/r4_ab7_cfg5.txt
