#!/bin/bash
# rocprofv3 --pmc passes over the bench's -vcf leg (one pass per counter set), per-kernel means printed:
#   scripts/pmc_vcf.sh <tag> <kernel-name regex> [counter sets...]
tag=$1; pat=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; ulimit -c 0
i=0
for set in "$@"; do
  i=$((i + 1))
  timeout 800 rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -o p -- python3 bench.py --steps 1 --warmup 1 --cpu-pairs 0 --vcf-reduce 1 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0 > $out/pmc$i.log 2>&1 < /dev/null
  f=$(find $out/pmc$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$pat" <<'P'
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if not pat.search(k):
        continue
    by[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in by.items():
    for c, v in cs.items():
        print(f"{k[:40]:40s} {c:24s} launches {len(v):3d}  mean {sum(v)/len(v):.4g}  max {max(v):.4g}")
P
  rm -rf $out/pmc$i
done
