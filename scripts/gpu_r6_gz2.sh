#!/bin/bash
# round 6: the line splitter cuts its lines where the feeder put them (no second copy of the text) — the input-side tests, then the .gz rates at 1 M and 4 M pairs
mkdir -p gpurun_out/r6g; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
timeout 600 python3 -m pytest tests -m gpu -x -q -k "input_side or plain_gz or bgzf or cli_sam_and_vcf or native_cli or golden or sharded_run" > gpurun_out/r6g/tests.log 2>&1 < /dev/null
tail -3 gpurun_out/r6g/tests.log
timeout 400 python3 scripts/gz_rate.py > gpurun_out/r6g/gz_rate.json 2> gpurun_out/r6g/gz_rate.err < /dev/null; cat gpurun_out/r6g/gz_rate.json
timeout 700 python3 scripts/gz_rate.py --pairs 4000000 --only-gz 1 > gpurun_out/r6g/gz_rate_4m.json 2> gpurun_out/r6g/gz_rate_4m.err < /dev/null; cat gpurun_out/r6g/gz_rate_4m.json
