#!/bin/bash
# round 6: the default run's timed region (12 steps, 2 warm-up) under rocprofv3 --kernel-trace --stats, and the default bench line a second time (spread)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6z; ulimit -c 0
out=gpurun_out/r6z/kt_default
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 bench.py --steps 12 --warmup 2 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0 > $out.log 2>&1 < /dev/null
f=$(find $out -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r6z/rocprofv3_kernel_stats_default_steps.csv; head -12 "$f" | cut -c1-160
grep -a "^{" $out.log | tail -1 | cut -c1-400
rm -rf $out
python3 bench.py > gpurun_out/r6z/bench_default_2.json 2> gpurun_out/r6z/bench_default_2.err < /dev/null
tail -c 600 gpurun_out/r6z/bench_default_2.json; python3 -c "
import json;o=json.loads(open('gpurun_out/r6z/bench_default_2.json').read().strip().splitlines()[-1]);print('second default run', o['value'], o['ms_per_step'], [ (c['ms_per_step']) for c in o['other_configs']])"
