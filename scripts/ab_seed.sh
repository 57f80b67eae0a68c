#!/bin/bash
# A/B of the seeding walk's forms on the bench workload (same records: the checksums say so)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 1100 -p no:cacheprovider -x -k "test_sam_equals_reference or large_batch_machinery or full_size" 2>&1 | tail -4
timeout 900 python scripts/ab_env.py --steps 5 --rounds 2 "" "MCX_SEED_WIDE64=1" > gpurun_out/ab_seed.txt 2> gpurun_out/ab_seed.err
cut -c1-200 gpurun_out/ab_seed.txt; python - <<'P'
import json
for l in open("gpurun_out/ab_seed.txt"):
    o = json.loads(l); print(o["variant"] or "default", o["ms_per_step"], o["stage_ms"]["seed"], o["checksum_records"], o["checksum_cigars"])
P
