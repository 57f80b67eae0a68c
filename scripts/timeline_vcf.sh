#!/bin/bash
# the -vcf leg of bench.py as a timeline: the kernels of one batch's period in the steady state (mapping of batch k+1, bookkeeping of batch k beside it)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; out=gpurun_out/timeline_vcf; mkdir -p $out; ulimit -c 0
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 6 --warmup 1 --cpu-pairs 0 --vcf-reduce 1 --pcie-steps 0 --second-genome 0 --other-configs 0 --file-steps 0 > $out/kt.log 2>&1 < /dev/null
python3 - $out <<'P'
import csv, glob, sys, os
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "kt", "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mcx::", "")[:40]))
ev.sort()
acc = [i for i, e in enumerate(ev) if e[3].startswith("k_prof_accum")]
i0 = acc[-3]  # the last run's batches: the third from the end is in the middle of the timed sequence
# time zero: the k_pack_reads before it... the mapping that runs beside this bookkeeping starts at the k_pack_reads just before k_prof_keys
t0 = ev[i0][0] - 3_000_000
with open(os.path.join(d, "timeline.txt"), "w") as out:
    for s, e, q, name in ev:
        if s < t0 or s > t0 + 44_000_000 or e - s < 60_000: continue
        out.write("%8.2f %8.2f %7.2f %-4s %s\n" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, name))
print(open(os.path.join(d, "timeline.txt")).read()[:14000])
P
rm -rf $out/kt
