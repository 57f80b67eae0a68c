#!/bin/bash
# round 6: mate rescue — the parity subset and the two timings that went with every change of k_rescue_eval (the seeds by all lanes: kept; the 8-mer table that picked the diagonals, whose A/B switch MCX_RESCUE_SCAN_ALL this script still sets: not kept, the switch is gone with it — DESIGN.md §3)
mkdir -p gpurun_out/r6r; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
timeout 1200 python3 -m pytest tests -m gpu -x -q -k "golden or sam_equals_reference or config5 or large_batch or fuzz_rounds or rescue or config2 or fresh_seeded or long_cigars or ragged or sharded" > gpurun_out/r6r/parity.log 2>&1 < /dev/null
tail -3 gpurun_out/r6r/parity.log
common="--cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0"
cfg5="--steps 4 --warmup 2 --rlen 250 --ins 0.025 --dele 0.025 $common"
for n in table all table all; do
  if [ $n = all ]; then export MCX_RESCUE_SCAN_ALL=1; else unset MCX_RESCUE_SCAN_ALL; fi
  timeout 300 python3 bench.py $cfg5 --alg nw --detail-tag r_$n > gpurun_out/r6r/c5_$n.json 2> gpurun_out/r6r/c5_$n.err < /dev/null
  python3 -c "
import json;o=json.loads(open('gpurun_out/r6r/c5_$n.json').read().strip().splitlines()[-1]);print('cfg5 $n',o['value'],o['ms_per_step'],o['stage_ms_per_step'])"
  timeout 300 python3 bench.py --steps 8 --warmup 3 $common --detail-tag rh_$n > gpurun_out/r6r/head_$n.json 2> gpurun_out/r6r/head_$n.err < /dev/null
  python3 -c "
import json;o=json.loads(open('gpurun_out/r6r/head_$n.json').read().strip().splitlines()[-1]);print('head $n',o['value'],o['ms_per_step'])"
done
