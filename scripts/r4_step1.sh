#!/bin/bash
# round 4, first evidence run: DP problem-size histograms (configs 5 and 3), a full bench batch against the compiled reference, the new tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
min="--cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0"
MCX_DP_HIST=1 timeout 600 python bench.py --steps 1 --warmup 1 $min --rlen 250 --ins 0.025 --dele 0.025 --alg nw > gpurun_out/dp_hist_cfg5.json 2> gpurun_out/dp_hist_cfg5.txt
MCX_DP_HIST=1 timeout 600 python bench.py --steps 1 --warmup 1 $min > gpurun_out/dp_hist_cfg3.json 2> gpurun_out/dp_hist_cfg3.txt
timeout 1500 python scripts/full_batch_parity.py --out gpurun_out/full_batch_parity.json > gpurun_out/full_batch_parity.log 2>&1
tail -c 1500 gpurun_out/full_batch_parity.log
timeout 900 python -m pytest tests -m gpu -q -x -k "cli_builds or settle or packed_host or smoke or golden_sam or sam_equals" -p no:cacheprovider 2>&1 | tail -5
