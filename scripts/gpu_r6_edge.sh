#!/bin/bash
# round 6: one edge word per row for a lane's two problems — parity subset, config 5 (nw and ksw2) and the headline, kernel trace + HBM bytes of the DP kernels
mkdir -p gpurun_out/r6e; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; ulimit -c 0
timeout 1000 python3 -m pytest tests -m gpu -x -q -k "${KSEL:-config5 or large_batch or packed_host_boundary or fresh_seeded or sam_equals_reference or extend or long_cigars or ragged or ksw2 or config2 or fuzz}" > gpurun_out/r6e/parity.log 2>&1 < /dev/null
tail -5 gpurun_out/r6e/parity.log
common="--cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0"
cfg5="--steps 2 --warmup 1 --rlen 250 --ins 0.025 --dele 0.025 $common"
timeout 400 python3 bench.py $cfg5 --alg nw --detail-tag e_cfg5_nw > gpurun_out/r6e/cfg5_nw.json 2> gpurun_out/r6e/cfg5_nw.err < /dev/null
timeout 400 python3 bench.py $cfg5 --alg ksw2 --detail-tag e_cfg5_ksw2 > gpurun_out/r6e/cfg5_ksw2.json 2> gpurun_out/r6e/cfg5_ksw2.err < /dev/null
timeout 400 python3 bench.py --steps 6 --warmup 2 $common --detail-tag e_head > gpurun_out/r6e/head.json 2> gpurun_out/r6e/head.err < /dev/null
out=gpurun_out/r6e/x
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py $cfg5 --alg nw > $out.kt.log 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc1 -o p -- python3 bench.py $cfg5 --alg nw > $out.pmc1.log 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc2 -o p -- python3 bench.py $cfg5 --alg nw > $out.pmc2.log 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out/pmc3 -o p -- python3 bench.py $cfg5 --alg nw > $out.pmc3.log 2>&1 < /dev/null
python3 scripts/summarize_profile.py --trace $out/kt --pmc $out/pmc1 $out/pmc2 $out/pmc3 --reads-per-launch 8000000 --batches 3 --out $out.summary.json --command "scripts/gpu_r6_edge.sh" < /dev/null
rm -rf $out
python3 - <<'P'
import json
for n in ("cfg5_nw","cfg5_ksw2","head"):
    try:
        o=json.loads(open(f"gpurun_out/r6e/{n}.json").read().strip().splitlines()[-1])
        print(n, o["value"], o["ms_per_step"], o.get("stage_ms_per_step"), o["roofline"].get("gcups"))
    except Exception as e:
        print(n, "failed", e)
s=json.load(open("gpurun_out/r6e/x.summary.json"))
for k,v in sorted(s["kernel_trace"].items()):
    if k.startswith("k_dp"): print("  ", k, v)
for k,p in s["pmc"].items():
    if k.startswith("k_dp_lane"): print("  ", k, {c:(x_.get("total"), x_.get("launches")) for c,x_ in p.items()})
P
