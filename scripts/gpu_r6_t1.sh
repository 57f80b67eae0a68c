#!/bin/bash
# round 6: config 5's large tier in one pass (MCX_TIER1_GB=48) against two (24)
mkdir -p gpurun_out/r6t; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
common="--cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0"
cfg5="--steps 3 --warmup 1 --rlen 250 --ins 0.025 --dele 0.025 $common"
for gb in 24 48 40; do
  MCX_TIER1_GB=$gb timeout 400 python3 bench.py $cfg5 --alg nw --detail-tag t1_$gb > gpurun_out/r6t/cfg5_$gb.json 2> gpurun_out/r6t/cfg5_$gb.err < /dev/null
done
MCX_TIER1_GB=48 timeout 600 python3 -m pytest tests -m gpu -x -q -k "config5 or large_batch" > gpurun_out/r6t/parity.log 2>&1 < /dev/null
tail -3 gpurun_out/r6t/parity.log
python3 - <<'P'
import json
for n in ("24","48","40"):
    try:
        o=json.loads(open(f"gpurun_out/r6t/cfg5_{n}.json").read().strip().splitlines()[-1])
        print(n, o["value"], o["ms_per_step"], o.get("stage_ms_per_step"), o.get("tier1_pairs"))
    except Exception as e:
        print(n, "failed", e)
P
