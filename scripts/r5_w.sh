#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "host_without_torch or bookkeeping_behind or packed_host" 2>&1 | tail -12 | cut -c1-300
