#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 900 python scripts/ab_env.py --steps 3 "" "" > gpurun_out/r4_ab7_cfg3.txt 2>/dev/null; cat gpurun_out/r4_ab7_cfg3.txt
MCX_LIB=$PWD/mapcaller_amd/libmcx_b5.so timeout 600 python scripts/ab_env.py --steps 3 "" "" 2>/dev/null | sed "s/^/b5 /"
timeout 3000 python -m pytest tests -m gpu -q --timeout 2400 -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/r4_pytest_full.log
tail -6 gpurun_out/r4_pytest_full.log
