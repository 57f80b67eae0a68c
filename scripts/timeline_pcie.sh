#!/bin/bash
# The boundary-inclusive leg of bench.py as a timeline: kernels and memory copies of two steps in the middle of the sequence (rocprofv3
# --kernel-trace --memory-copy-trace; no counters).  gpurun -- 'bash scripts/timeline_pcie.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; out=gpurun_out/timeline_pcie; mkdir -p $out; ulimit -c 0
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 2 --warmup 1 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 6 --second-genome 0 --other-configs 0 --file-steps 0 > $out/kt.log 2>&1 < /dev/null
grep -E "^\{" $out/kt.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('value_pcie_inclusive'))"
python3 - $out <<'P'
import csv, glob, sys, os
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "kt", "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]))
for f in glob.glob(os.path.join(d, "kt", "**", "*_memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", r.get("Name", "?")) + " " + r.get("Bytes", r.get("Size", "?"))))
ev.sort()
seeds = [i for i, e in enumerate(ev) if e[3].startswith("k_seed") and e[1] - e[0] > 2_000_000]
# the boundary leg's steps are the last six full-batch k_seed launches: take the fourth of them as time zero, print two steps
i0 = seeds[-3]
t0 = ev[i0][0]
with open(os.path.join(d, "timeline.txt"), "w") as out:
    for s, e, q, name in ev:
        if s < t0 - 3_000_000 or s > t0 + 45_000_000 or e - s < 40_000:
            continue
        out.write("%8.2f %8.2f %7.2f %-5s %s\n" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, name))
print(open(os.path.join(d, "timeline.txt")).read()[:12000])
P
rm -rf $out/kt
