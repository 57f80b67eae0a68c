#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
run() { tag=$1; shift; env "$@" python bench.py --steps 4 --warmup 1 --other-configs 0 --second-genome 0 --cpu-pairs 0 --vcf-reduce 0 --file-steps 0 --pcie-steps 12 > gpurun_out/r5_n_$tag.json 2> gpurun_out/r5_n_$tag.err
python - $tag <<'P'
import json, sys
l=[x for x in open('gpurun_out/r5_n_%s.json' % sys.argv[1]) if x.startswith('{')][-1]
d=json.loads(l); p=d.get('value_pcie_inclusive',{})
print(sys.argv[1], d['ms_per_step'], 'pcie', p.get('ms_per_step'))
P
}
run base A=1
run in16 MCX_STREAM_KERNEL_COPY=in MCX_COPY_BLOCKS=16
run in64 MCX_STREAM_KERNEL_COPY=in MCX_COPY_BLOCKS=64
run in256 MCX_STREAM_KERNEL_COPY=in MCX_COPY_BLOCKS=256
