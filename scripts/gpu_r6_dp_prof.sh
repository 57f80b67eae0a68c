#!/bin/bash
# round 6: where the DP stage of config 5 goes — kernel trace of the one- and the two-problems-per-lane forms, issue / wait counters of both
mkdir -p gpurun_out/r6c; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; ulimit -c 0
args="--steps 2 --warmup 1 --rlen 250 --ins 0.025 --dele 0.025 --alg nw --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0"
for x1 in 0 1; do
  if [ $x1 = 1 ]; then export MCX_DP_X1=1; else unset MCX_DP_X1; fi
  out=gpurun_out/r6c/x1_$x1
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py $args > $out.kt.log 2>&1 < /dev/null
  timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc1 -o p -- python3 bench.py $args > $out.pmc1.log 2>&1 < /dev/null
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc2 -o p -- python3 bench.py $args > $out.pmc2.log 2>&1 < /dev/null
  python3 scripts/summarize_profile.py --trace $out/kt --pmc $out/pmc1 $out/pmc2 --reads-per-launch 8000000 --batches 3 --out $out.summary.json --command "scripts/gpu_r6_dp_prof.sh MCX_DP_X1=$x1" < /dev/null
  find $out/kt -name "*kernel_stats.csv" -exec cp {} $out.kernel_stats.csv \;
  rm -rf $out
done
python3 - <<'P'
import json
for x in (0,1):
    s=json.load(open(f"gpurun_out/r6c/x1_{x}.summary.json"))
    print("MCX_DP_X1 =", x)
    for k,v in sorted(s["kernel_trace"].items()):
        if k.startswith("k_dp"): print("  ", k, v)
    for k,p in s["pmc"].items():
        if k.startswith("k_dp_lane"):
            print("  ", k, {c:(x_.get("total"), x_.get("launches")) for c,x_ in p.items()})
P
