#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; ulimit -c 0
run() { tag=$1; shift; env MCX_TIMING=1 "$@" python bench.py --steps 2 --warmup 1 --other-configs 0 --second-genome 0 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 $EXTRA > gpurun_out/r5_r_$tag.json 2> gpurun_out/r5_r_$tag.err
echo "== $tag"
python - $tag <<'P'
import json, sys
l=[x for x in open('gpurun_out/r5_r_%s.json' % sys.argv[1]) if x.startswith('{')][-1]
d=json.loads(l); f=d.get('value_file_to_file',{})
print('files', f.get('value'), f.get('seconds'), 'no sam', f.get('without_sam_output',{}).get('value'), f.get('without_sam_output',{}).get('seconds'))
P
grep "busy seconds\|wall seconds" gpurun_out/r5_r_$tag.err | tail -4 | cut -c1-260
grep nr_throttled /sys/fs/cgroup/cpu.stat
}
EXTRA="--file-threads 6" run t6 A=1
EXTRA="--file-threads 8" run t8 A=1
EXTRA="--file-threads 12" run t12 A=1
EXTRA="--file-threads 24" run t24 A=1
