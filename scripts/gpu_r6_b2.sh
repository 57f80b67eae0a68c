#!/bin/bash
# round 6: config 5 with nothing grown, with the growths on demand (default), and with both sizes from the start — one box
mkdir -p gpurun_out/r6b; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
common="--cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0"
cfg5="--steps 4 --warmup 2 --rlen 250 --ins 0.025 --dele 0.025 $common"
run() { # name, env...
  n=$1; shift
  env "$@" MCX_ALLOC_LOG=1 timeout 400 python3 bench.py $cfg5 --alg nw --detail-tag g_$n > gpurun_out/r6b/g_$n.json 2> gpurun_out/r6b/g_$n.err < /dev/null
  python3 -c "
import json;o=json.loads(open('gpurun_out/r6b/g_$n.json').read().strip().splitlines()[-1]);print('$n',o['value'],o['ms_per_step'],o['stage_ms_per_step']['dp'])"
  grep -a -c "grow" gpurun_out/r6b/g_$n.err
}
for rep in 1 2; do
run none MCX_NO_TIER1_GROW=1
run default X=1
run start MCX_TIER1_GB=50 MCX_DP_BLOCKS1=9216
done
