#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
MCX_TIMING=1 timeout 1500 python scripts/determinism_vcf.py > gpurun_out/r5_determinism_vcf.json 2> gpurun_out/r5_determinism_vcf.err; tail -c 500 gpurun_out/r5_determinism_vcf.json; echo; grep -c "queued behind" gpurun_out/r5_determinism_vcf.err; grep "no room in HBM" gpurun_out/r5_determinism_vcf.err | head -2

