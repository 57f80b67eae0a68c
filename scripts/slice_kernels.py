"""From a rocprofv3 kernel trace of the bench's -vcf leg: the kernels of the last slice mapped without the bookkeeping and of the
last slice mapped with it, summed by name (ms), plus the wall span of each slice on the GPU.
    python scripts/slice_kernels.py <kernel_trace.csv>"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
starts = [i for i, r in enumerate(rows) if name(r).startswith("k_max_read_len")]
slices = []
for a, b in zip(starts, starts[1:] + [len(rows)]):
    seg = rows[a:b]
    # (a slice ends with its last own kernel: what follows the final slice belongs to the settle / the variant caller)
    last = max((i for i, r in enumerate(seg) if name(r).startswith(("k_finish", "k_prof_disc", "mcx::k_prof_disc", "k_chunk"))), default=len(seg) - 1)
    seg = seg[: last + 1]
    slices.append((any("k_prof_accum" in name(r) for r in seg), seg))
for with_prof in (False, True):
    sel = [s for p, s in slices if p == with_prof]
    if not sel:
        continue
    seg = sel[-1]
    by = collections.Counter()
    for r in seg:
        by[name(r)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
    print(f"--- slice {'with' if with_prof else 'without'} the profile: {len(seg)} launches, span {span:.2f} ms, kernels {sum(by.values()):.2f} ms")
    for k, v in by.most_common(22):
        print(f"   {k:50s} {v:8.3f}")
    # where the GPU waited for the host inside the slice
    end = int(seg[0]["End_Timestamp"])
    gaps = []
    for a, b in zip(seg, seg[1:]):
        end = max(end, int(a["End_Timestamp"]))
        g = (int(b["Start_Timestamp"]) - end) / 1e6
        if g > 0.15:
            gaps.append((g, name(a), name(b)))
    print(f"   idle gaps > 0.15 ms: {sum(g for g, _, _ in gaps):.2f} ms in {len(gaps)}")
    for g, a, b in sorted(gaps, reverse=True)[:8]:
        print(f"      {g:7.3f} ms between {a} and {b}")
# the time from one slice's first kernel to the next one's
t = [int(rows[i]["Start_Timestamp"]) for i in starts]
print("slice starts apart (ms):", " ".join(f"{(b - a) / 1e6:.1f}" for a, b in zip(t, t[1:])))
