#!/bin/bash
# Collects a round's rocprofv3 evidence on the GPU box and leaves only small summaries in gpurun_out/<tag>/:
#   scripts/collect_profile.sh <tag> [human|uniform] [extra bench.py arguments...]
# kernel trace + stats of bench.py's timed workload, then one rocprofv3 --pmc pass per counter set (separate runs, as the
# guide prescribes; FETCH_SIZE and WRITE_SIZE do not fit one pass), condensed by scripts/summarize_profile.py.  Every command reads stdin from /dev/null.
# READS_PER_LAUNCH (default 8000000): reads one full-batch launch maps, for the per-read figures of the summary.
# Extra arguments select another workload (BASELINE configs 5 and 2: scripts/profile_configs.sh).
tag=${1:-prof}
genome=${2:-human}
shift; shift
extra="$*"
rpl=${READS_PER_LAUNCH:-8000000}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
args="--warmup 1 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0 --genome $genome $extra"
MCX_TIMING=1 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 3 $args > $out/kt.log 2>&1 < /dev/null
grep -E "^\[run_pairs\]|^\[mcx" $out/kt.log | head -40 > $out/job_classes.txt
grep -E "^\{" $out/kt.log | tail -1 > $out/bench_line.json
i=0
for set in ${PMC_SETS:+"$PMC_SETS"} "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i + 1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -o p -- python3 bench.py --steps 2 $args > $out/pmc$i.log 2>&1 < /dev/null
done
python3 scripts/summarize_profile.py --trace $out/kt --pmc $out/pmc* --reads-per-launch $rpl --batches 3 --out $out/summary.json \
  --command "scripts/collect_profile.sh $tag $genome $extra: rocprofv3 --kernel-trace --stats | --pmc <set> (one pass per set) -- python3 bench.py --steps 3|2 $args ($rpl reads per launch)" < /dev/null
find $out/kt -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/kt $out/pmc[0-9]
ls -la $out
