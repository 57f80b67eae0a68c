#!/bin/bash
# round 4, second run: the lane DP and the straight-line path on the GPU — parity tests first, then A/B timings with record checksums
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -q -x -k "sam_equals or extend or degenerate or long_cigars or cli_sam or overlapped or packed_host or full_size or large_batch or config5 or config2 or smoke" -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/r4_step2_pytest.log
tail -5 gpurun_out/r4_step2_pytest.log
timeout 900 python scripts/ab_env.py --steps 3 "" "MCX_NO_SIMPLE=1" "MCX_DP_BY_WAVE=1" "MCX_NO_SIMPLE=1 MCX_DP_BY_WAVE=1" "" > gpurun_out/r4_ab_cfg3.txt 2> gpurun_out/r4_ab_cfg3.err
cat gpurun_out/r4_ab_cfg3.txt
timeout 900 python scripts/ab_env.py --steps 2 --rlen 250 --ins 0.025 --dele 0.025 --alg nw "" "MCX_DP_NO_SORT=1" "MCX_DP_BY_WAVE=1" "" > gpurun_out/r4_ab_cfg5.txt 2> gpurun_out/r4_ab_cfg5.err
cat gpurun_out/r4_ab_cfg5.txt
tail -3 gpurun_out/r4_ab_cfg3.err gpurun_out/r4_ab_cfg5.err
