source scripts/gpu_exp.sh
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "test_sam_equals_reference or alignment_profile or fresh_seeded or degenerate or ragged or bwt_search" 2>&1 | tail -5 | cut -c1-300
run b6 base A=1
run b2 base MCX_SEED_FM_BUDGET=2
run b4 base MCX_SEED_FM_BUDGET=4
run b12 base MCX_SEED_FM_BUDGET=12
run b1000 base MCX_SEED_FM_BUDGET=1000
GENOME=uniform run u_b6 base A=1
GENOME=uniform run u_b1000 base MCX_SEED_FM_BUDGET=1000
