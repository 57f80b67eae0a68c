#!/bin/bash
# round 6, last: the whole GPU suite with its slowest tests, the smoke entry, the default bench line as the driver runs it (no flags), timed
mkdir -p gpurun_out/r6z; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
( time timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/r6z/gpu_suite.log 2>&1 < /dev/null
tail -22 gpurun_out/r6z/gpu_suite.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6z/smoke.log 2>&1; tail -1 gpurun_out/r6z/smoke.log
SECONDS=0
python3 bench.py > gpurun_out/r6z/bench_default.json 2> gpurun_out/r6z/bench_default.err < /dev/null
echo "default bench: $SECONDS s"; tail -c 3900 gpurun_out/r6z/bench_default.json
cp gpurun_out/bench_detail.json gpurun_out/r6z/bench_detail.json; cp gpurun_out/bench_detail_cfg5.json gpurun_out/bench_detail_cfg2.json gpurun_out/bench_detail_other_genome.json gpurun_out/r6z/ 2>/dev/null
