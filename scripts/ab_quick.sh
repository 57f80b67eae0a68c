#!/bin/bash
# one variant list through scripts/ab_env.py on the bench workload: ms per step, stage times, checksums
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 900 python scripts/ab_env.py --steps 5 "$@" > gpurun_out/ab_quick.txt 2> gpurun_out/ab_quick.err
python - <<'P'
import json
for l in open("gpurun_out/ab_quick.txt"):
    o = json.loads(l); print(o["variant"] or "default", o["ms_per_step"], o["stage_ms"], o["checksum_records"], o["checksum_cigars"])
P
