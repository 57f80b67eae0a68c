source scripts/gpu_exp.sh
run base base A=1
run noscan noscan A=1
run caps3 base MCX_FAST_CAPS=3,11
run caps4s base MCX_FAST_CAPS=4,13
run nofast base MCX_NO_FAST=1
GENOME=uniform run u_base base A=1
GENOME=uniform run u_caps3 base MCX_FAST_CAPS=3,11
GENOME=uniform run u_nofast base MCX_NO_FAST=1
cp mapcaller_amd/libmcx_base.so mapcaller_amd/libmcx.so
