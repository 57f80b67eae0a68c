#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "seeding_ahead or overlapped or test_sam_equals_reference or small_batches or fresh_seeded or sharded_run" 2>&1 | tail -6 | cut -c1-400
bash scripts/gpu_bench_quick.sh p
echo "--- no seed ahead"
MCX_NO_SEED_AHEAD=1 bash scripts/gpu_bench_quick.sh q
