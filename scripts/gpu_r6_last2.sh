#!/bin/bash
# round 6, the very last: the default bench line as the driver runs it (no flags) with this build's PMC summaries in place, then the final fuzz rounds
mkdir -p gpurun_out/r6z; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
SECONDS=0
python3 bench.py > gpurun_out/r6z/bench_default.json 2> gpurun_out/r6z/bench_default.err < /dev/null
echo "default bench: $SECONDS s"; tail -c 3900 gpurun_out/r6z/bench_default.json
cp gpurun_out/bench_detail.json gpurun_out/r6z/bench_detail.json; cp gpurun_out/bench_detail_cfg5.json gpurun_out/bench_detail_cfg2.json gpurun_out/bench_detail_other_genome.json gpurun_out/r6z/ 2>/dev/null
bash scripts/fuzz_r6_final.sh > gpurun_out/r6z/fuzz.log 2>&1; cat gpurun_out/fuzz_r6_final.txt
