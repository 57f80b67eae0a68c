#!/bin/bash
# after a change of the per-pair stages: the parity tests that cover them, then the step's time and checksums
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 1400 -p no:cacheprovider -x -k "test_sam_equals_reference or avgdist or large_batch_machinery or full_size or keeps_its_shape or sharded_run or eight_shards or fresh_seeded" 2>&1 | tail -4
bash scripts/ab_quick.sh "$@"
