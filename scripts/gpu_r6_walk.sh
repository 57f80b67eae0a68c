#!/bin/bash
# round 6: what the tracebacks of the lane DP cost at config 5 — the product against a build whose lane kernels leave the walks out (scripts/probe/libmcx_nowalk.so,
# -DMCX_DBG_SKIP_WALK: not alignments, a timing only)
# The build (here, before gpurun; the .so travels with the snapshot and is not committed):
#   cd mapcaller_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMCX_DBG_SKIP_WALK -c mcx_pipeline.hip -o /tmp/p_nowalk.o && \
#   hipcc --offload-arch=gfx950 -shared -fPIC /tmp/p_nowalk.o mcx_index_build.o mcx_variants.o mcx_files.o mcx_host.o -o ../../scripts/probe/libmcx_nowalk.so -lz -lpthread
mkdir -p gpurun_out/r6w; cd "$GRAFT_REPO_ROOT"; ulimit -c 0
common="--cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0"
cfg5="--steps 3 --warmup 2 --rlen 250 --ins 0.025 --dele 0.025 $common"
for n in walk nowalk walk nowalk; do
  if [ $n = nowalk ]; then export MCX_LIB=$GRAFT_REPO_ROOT/scripts/probe/libmcx_nowalk.so; else unset MCX_LIB; fi
  timeout 300 python3 bench.py $cfg5 --alg nw --detail-tag w_$n > gpurun_out/r6w/$n.json 2> gpurun_out/r6w/$n.err < /dev/null
  python3 -c "
import json;o=json.loads(open('gpurun_out/r6w/$n.json').read().strip().splitlines()[-1]);print('$n',o['value'],o['ms_per_step'],o['stage_ms_per_step']['dp'])"
done
