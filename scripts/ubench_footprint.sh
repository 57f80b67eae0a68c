#!/bin/bash
# dependent random 64-byte reads against the size of the table they fall into (what the TLB reach does to the seeding walk's ceiling)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
for mib in 1024 8192 32768 65536 114688 196608; do timeout 300 tools/ubench_gather $mib 256 quick; done > gpurun_out/ubench_gather_footprint.txt 2>&1
cat gpurun_out/ubench_gather_footprint.txt
