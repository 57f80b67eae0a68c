#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out /tmp/fz; ulimit -c 0
python scripts/fuzz_parity.py --rounds 60 --seed 2027 2>&1 | grep -v ": ok" | cut -c1-250
echo "--- no codes variant"
cp mapcaller_amd/libmcx_nocodes.so mapcaller_amd/libmcx.so
for r in 59; do python scripts/fuzz_parity.py --rounds 60 --seed 2027 --only $r 2>&1 | tail -1; done
cp mapcaller_amd/libmcx_base.so mapcaller_amd/libmcx.so
