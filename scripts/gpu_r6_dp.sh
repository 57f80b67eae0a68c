#!/bin/bash
# round 6: the two-problems-per-lane DP on the GPU — parity subset first, then config 5 and the headline with and without it (MCX_DP_X1=1: one problem per lane)
mkdir -p gpurun_out/r6b; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
timeout 900 python3 -m pytest tests -m gpu -x -q -k "config5 or large_batch or packed_host_boundary or fresh_seeded or sam_equals_reference or extend or long_cigars or ragged or bench_launches" > gpurun_out/r6b/parity.log 2>&1 < /dev/null
tail -5 gpurun_out/r6b/parity.log
common="--cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0"
for x1 in 0 1; do
  if [ $x1 = 1 ]; then export MCX_DP_X1=1; else unset MCX_DP_X1; fi
  timeout 400 python3 bench.py --steps 2 --warmup 1 --rlen 250 --ins 0.025 --dele 0.025 --alg nw $common --detail-tag cfg5_x1_$x1 > gpurun_out/r6b/cfg5_x1_$x1.json 2> gpurun_out/r6b/cfg5_x1_$x1.err < /dev/null
  timeout 400 python3 bench.py --steps 6 --warmup 2 $common --detail-tag head_x1_$x1 > gpurun_out/r6b/head_x1_$x1.json 2> gpurun_out/r6b/head_x1_$x1.err < /dev/null
done
unset MCX_DP_X1
python3 - <<'P'
import json
for n in ("cfg5_x1_0","cfg5_x1_1","head_x1_0","head_x1_1"):
    try:
        o=json.loads(open(f"gpurun_out/r6b/{n}.json").read().strip().splitlines()[-1])
        print(n, o["value"], o["ms_per_step"], o["stage_ms_per_step"], o["roofline"].get("gcups"))
    except Exception as e:
        print(n, "failed", e)
P
cp gpurun_out/bench_detail_*x1*.json gpurun_out/r6b/ 2>/dev/null
