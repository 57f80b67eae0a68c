#!/bin/bash
# round 5: config 4's slice at full size against the compiled reference, then the default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
free -g | head -2 > gpurun_out/r5_cfg4_test.log; nproc >> gpurun_out/r5_cfg4_test.log; df -h /tmp | tail -1 >> gpurun_out/r5_cfg4_test.log
SECONDS=0
timeout 3000 python -m pytest tests -m gpu -q --timeout 2800 -p no:cacheprovider -x -s -k "config4" 2>&1 | tail -30 >> gpurun_out/r5_cfg4_test.log
echo "test: $SECONDS s" >> gpurun_out/r5_cfg4_test.log; tail -5 gpurun_out/r5_cfg4_test.log
if [ "$1" != "nobench" ]; then
SECONDS=0; python bench.py --steps 10 --warmup 2 > gpurun_out/r5_bench0.json 2> gpurun_out/r5_bench0.err; echo "bench: $SECONDS s"
tail -c 600 gpurun_out/r5_bench0.json
fi
