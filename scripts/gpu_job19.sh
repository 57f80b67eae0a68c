#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/j21; ulimit -c 0

timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/j21/kt -o kt -- python3 bench.py --steps 2 --warmup 1 --cpu-pairs 0 --vcf-reduce 0 --pcie-steps 0 --second-genome 0 > gpurun_out/j21/kt.log 2>&1 < /dev/null
f=$(find gpurun_out/j21/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last k_seed with big grid = start of last step
idx=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("k_seed") and int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"])>=4096*64]
i0=idx[-1]
t0=int(rows[i0]["Start_Timestamp"])
out=open("gpurun_out/j21/timeline.txt","w")
for r in rows[i0-2:]:
    s=(int(r["Start_Timestamp"])-t0)/1e6; e=(int(r["End_Timestamp"])-t0)/1e6
    if e-s<0.05: continue
    out.write("%8.2f %8.2f %7.2f q%s %s grid %s\n"%(s,e,e-s,r.get("Queue_Id","?"),r["Kernel_Name"][:50],r.get("Grid_Size_X",r.get("Grid_Size","?"))))
out.close()
print(open("gpurun_out/j21/timeline.txt").read()[:6000])
P
rm -rf gpurun_out/j21/kt
