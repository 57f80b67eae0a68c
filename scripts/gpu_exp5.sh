source scripts/gpu_exp.sh
GENOME=uniform run u_w7 base A=1
GENOME=uniform run u_w6 w6 A=1
GENOME=uniform run u_w5 w5 A=1
GENOME=uniform run u_w4 w4 A=1
run h_w7 base A=1
run h_w5 w5 A=1
run h_caps64 base MCX_TIER0_CAPS=64,12,96,1024
run h_caps64_16 base MCX_TIER0_CAPS=64,16,96,1024
cp mapcaller_amd/libmcx_base.so mapcaller_amd/libmcx.so
