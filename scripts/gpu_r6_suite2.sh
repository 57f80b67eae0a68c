#!/bin/bash
# round 6: the GPU suite again (durations), the .gz rates, and the headline / config 5 with every DP list on the lane kernels
mkdir -p gpurun_out/r6f; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
( time timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r6f/gpu_suite.log 2>&1 < /dev/null
tail -30 gpurun_out/r6f/gpu_suite.log
timeout 900 python3 scripts/gz_rate.py > gpurun_out/r6f/gz_rate.json 2> gpurun_out/r6f/gz_rate.err < /dev/null
tail -2 gpurun_out/r6f/gz_rate.json
common="--cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0"
for v in default lane_always; do
  if [ $v = lane_always ]; then export MCX_DP_LANE_ALWAYS=1; else unset MCX_DP_LANE_ALWAYS; fi
  timeout 400 python3 bench.py --steps 6 --warmup 2 $common --detail-tag head_$v > gpurun_out/r6f/head_$v.json 2> gpurun_out/r6f/head_$v.err < /dev/null
done
unset MCX_DP_LANE_ALWAYS
python3 - <<'P'
import json
for n in ("head_default","head_lane_always"):
    try:
        o=json.loads(open(f"gpurun_out/r6f/{n}.json").read().strip().splitlines()[-1])
        print(n, o["value"], o["ms_per_step"], o["stage_ms_per_step"])
    except Exception as e:
        print(n, "failed", e)
P
