#!/bin/bash
# gpurun -- 'bash scripts/probe/run_d2h_probe.sh': the copy probes (hipcc --offload-arch=gfx950 -O2 -o scripts/probe/d2h_probe scripts/probe/d2h_probe.hip first)
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/d2h_probe; mkdir -p $out; ulimit -c 0
echo "== C probe, the system's runtime"; timeout 120 scripts/probe/d2h_probe
echo "== python with torch"; timeout 300 python3 scripts/probe/d2h_probe.py 2>&1 | tail -6
AMD_LOG_LEVEL=4 timeout 300 python3 scripts/probe/d2h_probe.py > $out/py.log 2>&1
grep -iE "HSA Copy|Query copy" $out/py.log | sed -e 's/.*tid: [0-9a-fx]*\] //' | sed -e 's/dst=[0-9a-fx]*, src=[0-9a-fx]*, //; s/, wait_event.*//' | sort | uniq -c
