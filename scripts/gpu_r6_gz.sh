#!/bin/bash
# the .gz paths: tests and rates
mkdir -p gpurun_out/r6i; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
timeout 900 python3 -m pytest tests -m gpu -x -q -k "input_side or plain_gz or bgzf" > gpurun_out/r6i/gz_tests.log 2>&1 < /dev/null
tail -3 gpurun_out/r6i/gz_tests.log
for v in default serial_parser; do
  if [ $v = serial_parser ]; then export MCX_SERIAL_PARSER=1; else unset MCX_SERIAL_PARSER; fi
  timeout 900 python3 scripts/gz_rate.py > gpurun_out/r6i/gz_rate_$v.json 2> gpurun_out/r6i/gz_rate_$v.err < /dev/null
  echo $v; cat gpurun_out/r6i/gz_rate_$v.json
done
