#!/bin/bash
# round 6: the two-problems-per-lane DP with its rows prefetched and its walks windowed — parity subset, config 5 and headline A/B, kernel trace + wait counters
mkdir -p gpurun_out/r6d; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; ulimit -c 0
timeout 900 python3 -m pytest tests -m gpu -x -q -k "config5 or large_batch or packed_host_boundary or fresh_seeded or sam_equals_reference or extend or long_cigars or ragged or hbm_that_is_left or bench_launches" > gpurun_out/r6d/parity.log 2>&1 < /dev/null
tail -5 gpurun_out/r6d/parity.log
common="--cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0"
cfg5="--steps 2 --warmup 1 --rlen 250 --ins 0.025 --dele 0.025 --alg nw $common"
for x1 in 0 1; do
  if [ $x1 = 1 ]; then export MCX_DP_X1=1; else unset MCX_DP_X1; fi
  timeout 400 python3 bench.py $cfg5 --detail-tag cfg5_x1_$x1 > gpurun_out/r6d/cfg5_x1_$x1.json 2> gpurun_out/r6d/cfg5_x1_$x1.err < /dev/null
  timeout 400 python3 bench.py --steps 6 --warmup 2 $common --detail-tag head_x1_$x1 > gpurun_out/r6d/head_x1_$x1.json 2> gpurun_out/r6d/head_x1_$x1.err < /dev/null
done
unset MCX_DP_X1
out=gpurun_out/r6d/x2
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py $cfg5 > $out.kt.log 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc1 -o p -- python3 bench.py $cfg5 > $out.pmc1.log 2>&1 < /dev/null
python3 scripts/summarize_profile.py --trace $out/kt --pmc $out/pmc1 --reads-per-launch 8000000 --batches 3 --out $out.summary.json --command "scripts/gpu_r6_dp2.sh" < /dev/null
rm -rf $out
python3 - <<'P'
import json
for n in ("cfg5_x1_0","cfg5_x1_1","head_x1_0","head_x1_1"):
    try:
        o=json.loads(open(f"gpurun_out/r6d/{n}.json").read().strip().splitlines()[-1])
        print(n, o["value"], o["ms_per_step"], o["stage_ms_per_step"], o["roofline"].get("gcups"))
    except Exception as e:
        print(n, "failed", e)
s=json.load(open("gpurun_out/r6d/x2.summary.json"))
for k,v in sorted(s["kernel_trace"].items()):
    if k.startswith("k_dp"): print("  ", k, v)
for k,p in s["pmc"].items():
    if k.startswith("k_dp_lane"): print("  ", k, {c:(x_.get("total"), x_.get("launches")) for c,x_ in p.items()})
P
cp gpurun_out/bench_detail_*x1*.json gpurun_out/bench_ranks_*.txt gpurun_out/r6d/ 2>/dev/null
