#!/bin/bash
# round 6: the large tier's records grow on demand — the test, config 5 and the headline with the default settings
mkdir -p gpurun_out/r6t; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
timeout 900 python3 -m pytest tests -m gpu -x -q -k "scratch_grow or config5 or large_batch or hbm_that_is_left" > gpurun_out/r6t/parity2.log 2>&1 < /dev/null
tail -5 gpurun_out/r6t/parity2.log
common="--cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0"
MCX_ALLOC_LOG=1 timeout 400 python3 bench.py --steps 3 --warmup 1 --rlen 250 --ins 0.025 --dele 0.025 $common --alg nw --detail-tag t1_grow > gpurun_out/r6t/cfg5_grow.json 2> gpurun_out/r6t/cfg5_grow.err < /dev/null
grep -a "records grow" gpurun_out/r6t/cfg5_grow.err
MCX_ALLOC_LOG=1 timeout 400 python3 bench.py --steps 6 --warmup 2 $common --detail-tag t1_head > gpurun_out/r6t/head.json 2> gpurun_out/r6t/head.err < /dev/null
grep -a "records grow" gpurun_out/r6t/head.err
python3 - <<'P'
import json
for n in ("cfg5_grow","head"):
    try:
        o=json.loads(open(f"gpurun_out/r6t/{n}.json").read().strip().splitlines()[-1])
        print(n, o["value"], o["ms_per_step"], o.get("stage_ms_per_step"), o.get("tier1_pairs"))
    except Exception as e:
        print(n, "failed", e)
P
