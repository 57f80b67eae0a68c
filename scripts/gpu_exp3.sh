source scripts/gpu_exp.sh
run caps64_32 base MCX_TIER0_CAPS=64,32,128,1024
run caps96_48 base MCX_TIER0_CAPS=96,48,160,1536
run caps128_64 base MCX_TIER0_CAPS=128,64,192,2048
cp mapcaller_amd/libmcx_base.so mapcaller_amd/libmcx.so
