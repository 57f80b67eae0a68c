#!/bin/bash
# One step of bench.py as a timeline: which kernel ran when, on which hardware queue (rocprofv3 --kernel-trace).
# This is how the concurrency of the tiers is looked at:  gpurun -- 'bash scripts/timeline.sh [env assignments]'
# BENCH_ARGS="--rlen 250 ..." selects another workload, TAG=name another output directory (gpurun_out/timeline_<name>).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/timeline${TAG:+_$TAG}; ulimit -c 0
for kv in "$@"; do export "$kv"; done
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/timeline${TAG:+_$TAG}/kt -o kt -- python3 bench.py --steps 2 --warmup 1 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0 $BENCH_ARGS > gpurun_out/timeline${TAG:+_$TAG}/kt.log 2>&1 < /dev/null
f=$(find gpurun_out/timeline${TAG:+_$TAG}/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys, os
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_seed") and int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) >= 4096 * 64]
i0 = big[-1]
t0 = int(rows[i0]["Start_Timestamp"])
import os
tdir = "gpurun_out/timeline" + ("_" + os.environ["TAG"] if os.environ.get("TAG") else "")
with open(tdir + "/timeline.txt", "w") as out:
    for r in rows[max(0, i0 - 2):]:
        s = (int(r["Start_Timestamp"]) - t0) / 1e6
        e = (int(r["End_Timestamp"]) - t0) / 1e6
        if e - s < float(os.environ.get("MIN_MS", "0.05")):
            continue
        out.write("%8.2f %8.2f %7.2f q%s %s grid %s\n" % (s, e, e - s, r.get("Queue_Id", "?"), r["Kernel_Name"][:50], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
print(open(tdir + "/timeline.txt").read()[:9000])
P
rm -rf gpurun_out/timeline${TAG:+_$TAG}/kt
