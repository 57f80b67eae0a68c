source scripts/gpu_exp.sh
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "test_sam_equals_reference or fused_kernel or alignment_profile or vcf_equals or fresh_seeded or degenerate or ragged or long_cigars or maximum_read or overlapped or full_size" 2>&1 | tail -30 > gpurun_out/r2_pytest7.log; tail -6 gpurun_out/r2_pytest7.log | cut -c1-600
run base base A=1
GENOME=uniform run u_base base A=1
run fast base MCX_FAST=1
cp mapcaller_amd/libmcx_base.so mapcaller_amd/libmcx.so
bash scripts/gpu_bench_quick.sh g
