#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "test_sam_equals_reference or fused_kernel or alignment_profile or vcf_equals or fresh_seeded or degenerate or ragged or long_cigars or maximum_read or overlapped" 2>&1 | tail -30 > gpurun_out/r2_pytest6.log; tail -6 gpurun_out/r2_pytest6.log | cut -c1-600
bash scripts/gpu_bench_quick.sh f
