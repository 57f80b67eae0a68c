#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into the small summaries kept under profiles/.

  python scripts/summarize_profile.py --trace gpurun_out/prof_kt3 --pmc gpurun_out/prof3_* \
         --reads-per-launch 4000000 --out profiles/round1/summary_2gbp_final.json

--trace: a `rocprofv3 --kernel-trace --stats` directory (kernel_trace.csv) -> per kernel: launches,
         average / min / max duration, and the same over the full-batch launches only (grid of the
         largest launch), because the avgDist replay of the first batch adds a short launch.
--pmc:   `rocprofv3 --pmc ...` directories (counter_collection.csv) -> per kernel and counter, the
         mean over the full-batch launches.  FETCH_SIZE / WRITE_SIZE are reported in KiB by
         rocprofv3; bytes = value * 1024 (no x2: see the note written into the summary).
Only kernels of this package (name starts with k_) are kept.
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict


STREAMING = ("k_pack_reads", "k_unpack_reads")  # kernels whose reads are wide coalesced streams


def short(name):
    m = re.match(r"(?:void )?((?:mcx::)?k_[A-Za-z0-9_]+(?:<[^>(]*>)?)", name)
    return m.group(1) if m else None


def read_trace(d):
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                if k:
                    rows[k].append((int(r["Grid_Size_X"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out = {}
    for k, v in rows.items():
        gmax = max(g for g, _ in v)
        full = [t for g, t in v if g == gmax]
        # launches with the grid of the largest launch; for fixed-grid kernels fall back on duration
        if len(full) == len(v) and len(v) > 1:
            tmax = max(t for _, t in v)
            full = [t for _, t in v if t > 0.5 * tmax]
        alls = [t for _, t in v]
        out[k] = {"launches": len(alls), "avg_us": round(sum(alls) / len(alls) / 1e3, 1), "min_us": round(min(alls) / 1e3, 1),
                  "max_us": round(max(alls) / 1e3, 1), "full_batch_launches": len(full),
                  "full_batch_avg_us": round(sum(full) / len(full) / 1e3, 1)}
    return out


def read_pmc(dirs):
    vals = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            meta = {}
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    k = short(r["Kernel_Name"])
                    if not k:
                        continue
                    key = (k, r["Dispatch_Id"], r["Counter_Name"])
                    per_dispatch[key] += float(r["Counter_Value"])
                    meta[key] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            for (k, _, c), v in per_dispatch.items():
                vals[k][c].append((meta[(k, _, c)], v))
    out = {}
    for k, cs in vals.items():
        out[k] = {}
        for c, lst in cs.items():
            tmax = max(t for t, _ in lst)
            full = [v for t, v in lst if t > 0.5 * tmax]
            # total: over every launch of the run (a stage made of several launches of one kernel — the DP lists — is their sum)
            out[k][c] = {"launches": len(lst), "full_batch_mean": round(sum(full) / len(full), 2), "total": round(sum(v for _, v in lst), 2)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--reads-per-launch", type=int, default=0)
    ap.add_argument("--batches", type=int, default=0, help="batches the profiled command mapped (warm-up + steps): written into the summary, for per-step sums of the totals")
    ap.add_argument("--command", default="")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    s = {"command": a.command,
         "note": "FETCH_SIZE/WRITE_SIZE are KiB as rocprofv3 reports them. For the scattered 16-B-per-lane reads of k_seed, "
                 "FETCH_SIZE*1024 equals TCC_MISS_sum*64 within 1 %, i.e. 64-byte fabric requests counted at 64 B: the guide's "
                 "x2 correction (wide coalesced streams tallied as 128-B requests at 64 B) does not apply to this pattern; it is applied to "
                 "the streaming kernels (" + ", ".join(STREAMING) + ")."}
    try:  # the kernel sources the profiled build was made from (bench.py refuses a summary whose hash is not its own tree's)
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        s["kernel_source_sha"] = bench.kernel_source_sha()
    except Exception:
        pass
    if a.batches:
        s["batches_mapped_by_the_pmc_runs"] = a.batches
    if a.trace:
        s["kernel_trace"] = read_trace(a.trace)
    if a.pmc:
        s["pmc"] = read_pmc(a.pmc)
        if a.reads_per_launch:
            per = {}
            for k, cs in s["pmc"].items():
                e = {}
                if "FETCH_SIZE" in cs:
                    # wide coalesced streaming reads (16 B per lane, neighbouring lanes neighbouring addresses) are reported at half
                    # their bytes on gfx950 (MI355X_MICROARCH.md, HBM): doubled for the kernels that read that way
                    f = 2.0 if k in STREAMING else 1.0
                    e["hbm_read_bytes_per_launch"] = round(f * cs["FETCH_SIZE"]["full_batch_mean"] * 1024)
                    e["hbm_read_bytes_per_read"] = round(f * cs["FETCH_SIZE"]["full_batch_mean"] * 1024 / a.reads_per_launch, 1)
                    if f != 1.0:
                        e["fetch_size_correction"] = "x2 (wide coalesced streaming read)"
                if "WRITE_SIZE" in cs:
                    e["hbm_write_bytes_per_launch"] = round(cs["WRITE_SIZE"]["full_batch_mean"] * 1024)
                    e["hbm_write_bytes_per_read"] = round(cs["WRITE_SIZE"]["full_batch_mean"] * 1024 / a.reads_per_launch, 1)
                if e:
                    per[k] = e
            s["hbm_traffic"] = per
    with open(a.out, "w") as fh:
        json.dump(s, fh, indent=1, sort_keys=True)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
