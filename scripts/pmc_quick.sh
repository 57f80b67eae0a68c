#!/bin/bash
# rocprofv3 --pmc passes over scripts/ab_env.py (one pass per counter set), per-kernel sums of the biggest launches printed:
#   scripts/pmc_quick.sh <tag> <kernel-name regex> "<variant env>" [counter sets...]
tag=$1; pat=$2; variant=$3; shift 3
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; ulimit -c 0
for kv in $variant; do export "$kv"; done
i=0
for set in "$@"; do
  i=$((i + 1))
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -o p -- python3 scripts/ab_env.py --steps 1 "" > $out/pmc$i.log 2>&1 < /dev/null
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
    by[k.split("(")[0]][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
for k, cs in by.items():
    for c, v in cs.items():
        g = max(x[0] for x in v)
        big = [x[1] for x in v if x[0] == g]
        print(f"{k[:40]:40s} {c:24s} launches {len(big):2d} (grid {g})  mean {sum(big)/len(big):.4g}")
P
  rm -rf $out/pmc$i
done
