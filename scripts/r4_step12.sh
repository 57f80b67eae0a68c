#!/bin/bash
# the whole GPU suite on the current build, then the default bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 3000 python -m pytest tests -m gpu -q --timeout 2400 -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/r4_pytest_full2.log
tail -6 gpurun_out/r4_pytest_full2.log
SECONDS=0; python bench.py --steps 10 --warmup 2 --other-configs 0 --second-genome 0 > gpurun_out/r4_bench_mid.json 2> gpurun_out/r4_bench_mid.err; echo "bench: $SECONDS s"
python - <<'P'
import json
d = json.load(open("gpurun_out/r4_bench_mid.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["stage_ms_per_step"], d["value_pcie_inclusive"], str(d["value_file_to_file"])[:300])
P
