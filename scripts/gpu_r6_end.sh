#!/bin/bash
# round 6, towards the end: the .gz paths (tests + rates), then the default bench line as the driver runs it
mkdir -p gpurun_out/r6h; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
timeout 900 python3 -m pytest tests -m gpu -x -q -k "input_side or plain_gz or bgzf or cli_sam_and_vcf or sharded_run or native_cli or sam_equals_reference" > gpurun_out/r6h/gz_tests.log 2>&1 < /dev/null
tail -4 gpurun_out/r6h/gz_tests.log
timeout 900 python3 scripts/gz_rate.py > gpurun_out/r6h/gz_rate.json 2> gpurun_out/r6h/gz_rate.err < /dev/null
cat gpurun_out/r6h/gz_rate.json
SECONDS=0
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r6h/bench_default.json 2> gpurun_out/r6h/bench_default.err < /dev/null
echo "default bench: $SECONDS s"; tail -c 3800 gpurun_out/r6h/bench_default.json
cp gpurun_out/bench_detail.json gpurun_out/r6h/bench_detail.json; cp gpurun_out/bench_detail_cfg5.json gpurun_out/bench_detail_cfg2.json gpurun_out/bench_detail_other_genome.json gpurun_out/r6h/ 2>/dev/null
