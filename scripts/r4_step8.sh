#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 3000 python -m pytest tests -m gpu -q --timeout 2400 -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/r4_pytest_full.log
tail -6 gpurun_out/r4_pytest_full.log
timeout 900 python scripts/ab_env.py --steps 2 --rlen 250 --ins 0.025 --dele 0.025 --alg nw "" "MCX_TIER1_GB=8" "MCX_RESCUE_IN_LINE=1" > gpurun_out/r4_ab6_cfg5.txt 2> gpurun_out/r4_ab6_cfg5.err
cat gpurun_out/r4_ab6_cfg5.txt
for v in "" "MCX_NO_SUMS_CACHE=1"; do
  env $v MCX_BENCH_SHARE_GPU=1 MCX_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 timeout 900 python bench.py --gpus 2 --steps 4 --warmup 1 --genome-mbp 300 --contigs 8 --batch-pairs 1000000 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); print('two ranks sharing the GPU, $v:', o['value'], o['ms_per_step'], o['config']['multi_gpu_host_ms_per_step'], o['config']['multi_gpu'][-90:])"
done
