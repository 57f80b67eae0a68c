#!/bin/bash
# round 6: the 65-256-column list's scratch (how many wavefronts of k_dp_lane2 it lets run): 4 GB against 8 GB at config 5
mkdir -p gpurun_out/r6b; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
common="--cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0"
cfg5="--steps 4 --warmup 2 --rlen 250 --ins 0.025 --dele 0.025 $common"
for b in 4096 8192 4096 8192; do
  MCX_DP_BLOCKS1=$b timeout 400 python3 bench.py $cfg5 --alg nw --detail-tag b1_$b > gpurun_out/r6b/cfg5_$b.json 2> gpurun_out/r6b/cfg5_$b.err < /dev/null
  python3 -c "
import json;o=json.loads(open('gpurun_out/r6b/cfg5_$b.json').read().strip().splitlines()[-1]);print($b,o['value'],o['ms_per_step'],o['stage_ms_per_step']['dp'])"
done
