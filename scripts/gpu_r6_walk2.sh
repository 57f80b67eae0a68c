#!/bin/bash
# round 6: eight-row walk windows + scalar-base addressing of a lane's words — parity subset, config 5 (nw, ksw2), the headline
mkdir -p gpurun_out/r6w; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
timeout 1000 python3 -m pytest tests -m gpu -x -q -k "config5 or large_batch or packed_host_boundary or fresh_seeded or sam_equals_reference or extend or long_cigars or ragged or ksw2 or config2 or fuzz_rounds_equal or scratch_grow" > gpurun_out/r6w/parity.log 2>&1 < /dev/null
tail -3 gpurun_out/r6w/parity.log
common="--cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0"
cfg5="--steps 4 --warmup 2 --rlen 250 --ins 0.025 --dele 0.025 $common"
for n in nw ksw2 nw; do
  timeout 300 python3 bench.py $cfg5 --alg $n --detail-tag w2_$n > gpurun_out/r6w/c5_$n.json 2> gpurun_out/r6w/c5_$n.err < /dev/null
  python3 -c "
import json;o=json.loads(open('gpurun_out/r6w/c5_$n.json').read().strip().splitlines()[-1]);print('$n',o['value'],o['ms_per_step'],o['stage_ms_per_step']['dp'])"
done
timeout 300 python3 bench.py --steps 8 --warmup 3 $common --detail-tag w2_head > gpurun_out/r6w/head.json 2> gpurun_out/r6w/head.err < /dev/null
python3 -c "
import json;o=json.loads(open('gpurun_out/r6w/head.json').read().strip().splitlines()[-1]);print('head',o['value'],o['ms_per_step'],o['stage_ms_per_step'])"
