#!/bin/bash
# the .gz paths: tests and rates
mkdir -p gpurun_out/r6i; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
timeout 900 python3 -m pytest tests -m gpu -x -q -k "input_side or plain_gz or bgzf or cli_two_libraries or degenerate or ragged" > gpurun_out/r6i/gz_tests.log 2>&1 < /dev/null
tail -3 gpurun_out/r6i/gz_tests.log
for rep in 1 2; do
  timeout 900 python3 scripts/gz_rate.py > gpurun_out/r6i/gz_rate_$rep.json 2> gpurun_out/r6i/gz_rate_$rep.err < /dev/null
  cat gpurun_out/r6i/gz_rate_$rep.json
done
