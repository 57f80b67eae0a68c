#!/usr/bin/env python3
"""bench.py — reads/s of the seed-and-extend hot path on MI355X (BASELINE.json metric).

A *step* is one pass of the whole path (2-bit packing, FM-index seeding, SA resolution, clustering /
pairing / rescue, fragment construction, wavefront DP, scoring / MAPQ / CIGAR, avgDist replay)
over one batch of synthetic 150 bp paired-end reads that already sit in HBM; the results
(alignment records + CIGAR words) stay in HBM.  FASTQ parsing and SAM text are outside the timed
region.

Workload: GRCh38 itself cannot be obtained offline, so the genome is synthetic — uniform random
contigs with planted dispersed repeats — sized like GRCh38 by default (--genome-mbp 3100; the GPU
index builder handles genomes up to 4.29 Gbp) and indexed on the GPU by the product's own builder
inside this script (not timed).  Reads follow SURVEY.md §8d: 150 bp pairs, fragment N(500,50)
clipped to [300,800], 0.5 % substitutions, 0.1 % insertions, 0.1 % deletions per base.

One process per GPU (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE); every rank maps its own shard of
pairs against its own replica of the index: weak scaling, no collective on the data path.
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
# vector-instruction issue: 256 CUs x 4 SIMD-32, a wave64 instruction every 2 cycles at 2.4 GHz = 1.23 T wave-instructions/s = 78.6 T lane-operations/s
VALU_PEAK_TLANEOPS = 256 * 4 * 2.4e9 / 2 * 64 / 1e12
# vector instructions of the sweep's inner loop per cell, as compiled for gfx950.  Two problems per lane in 16-bit halves (k_dp_lane2, the default since round 6): a row of a strip
# is 283 (nw) / 454 (ksw2) vector instructions for its 2 x 16 cells; one problem per lane (k_dp_lane, MCX_DP_X1=1): 340 / 16 and 540 / 16
DP_OPS_PER_CELL = {"nw": 8.84, "ksw2": 14.19}
DP_OPS_PER_CELL_X1 = {"nw": 21.0, "ksw2": 34.0}
# what the packed recurrences themselves ask per cell — 17 (nw) / 26 (ksw2) v_pk_* instructions per pair of cells, mcx_dp_lane2.h —: the distance between this and the
# compiled count is the row's fetches and stores, the flag words' packing and the loop
DP_OPS_PER_CELL_MINIMAL = {"nw": 8.5, "ksw2": 13.0}
# random 16-byte gathers from an 8 GiB table, four lanes per 64-byte block: what the chip's L2 / fabric sustains in
# requests per second (tools/ubench_gather.hip, profiles/round1/ubench_gather_8GiB.txt: 47-48 G/s)
GATHER_CEILING_G_PER_S = 47.5
# the seeding walk's own pattern — every lane chases its own chain, ONE 16-byte load per 64-byte line — against the size of the table the
# reads fall into (tools/ubench_gather.hip mode 7, profiles/round4/ubench_gather_footprint.txt): 57 G lines/s in 1 GiB, 41 in 8 GiB,
# 37-38 from 32 to 192 GiB (the index is 110 GB)
WALK_MISS_CEILING_G_PER_S = 37.8
# committed rocprofv3 --pmc passes, by workload: (genome kind, Mbp, pairs per step, read length, alg, sub, ins, del, single-end)
PMC_SUMMARY = {
    ("human", 3100.0, 4_000_000, 150, "ksw2", 0.005, 0.001, 0.001, 0): "summary_human.json",
    ("uniform", 3100.0, 4_000_000, 150, "ksw2", 0.005, 0.001, 0.001, 0): "summary_uniform.json",
    ("human", 3100.0, 4_000_000, 250, "nw", 0.005, 0.025, 0.025, 0): "summary_cfg5.json",
    ("uniform", 4.6, 1_000_000, 100, "ksw2", 0.005, 0.001, 0.001, 1): "summary_cfg2.json",
}
PMC_ROUNDS = ("profiles/round6", "profiles/round5", "profiles/round4")  # the newest committed pass of a workload is the one that is read
# the stages the library times with HIP events on its own stream, and the kernel(s) each one is (tier 0 of a pass; the large tier's passes and the
# replay run the same kernels on other streams, beside them).  A stage of ONE kernel gives that kernel's live launch time.
STAGE_KERNELS = {"ms_encode": ["k_pack_reads"], "ms_seed": ["k_seed"],
                 "ms_simple": ["k_simple<{nw}, 1, false>", "k_simple_dp<{nw}>", "k_simple<{nw}, 2, false>"],  # collect, solve, replay
                 "ms_order": ["k_order_count", "k_order_place"], "ms_cluster": ["k_cluster"],
                 "ms_rescue": ["k_rescue_plan", "k_rescue_eval<2048>", "k_rescue_apply"], "ms_build": ["k_build"], "ms_finish": ["k_finish"]}
STEP_KERNEL_PREFIXES = ("k_pack_reads", "k_seed", "k_simple", "k_order_", "k_cluster", "k_rescue", "k_build", "k_dp_", "k_finish", "k_chunk_sums", "k_reduce_stats",
                        "k_check_est", "k_max_read_len", "k_fill_i32", "k_sa", "k_gather_pout")  # what a step launches (not the index builder's kernels)


def kernel_source_sha():
    """What the device code is made of, as one hash: the committed PMC summaries carry it (scripts/summarize_profile.py --source-sha), and a summary made
    from other sources than the ones this run was built from is not this run's traffic."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "mapcaller_amd", "csrc")
    host_only = {"mcx_pgz.h", "mcx_cpus.h", "mcx_variants_host.h"}  # (read by no kernel: the .gz reader, the CPU count, the variant caller's host passes)
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h"))):
        if os.path.basename(f) in host_only:
            continue
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_profile(args):
    """What the committed rocprofv3 --pmc passes of this same command measured per launch (counters cannot be read from
    inside the run): HBM bytes (FETCH_SIZE + WRITE_SIZE), L2 requests (TCC_HIT + TCC_MISS), vector instructions and wave
    cycles per kernel.  Only returned when the workload is one of those the passes profiled (scripts/collect_profile.sh,
    scripts/profile_configs.sh run bench.py with exactly these arguments)."""
    sig = (args.genome, float(args.genome_mbp), args.batch_pairs, args.rlen, args.alg, args.sub, args.ins, args.dele, int(bool(args.single_end)))
    name = PMC_SUMMARY.get(sig)
    path = next((os.path.join(r, name) for r in PMC_ROUNDS if name and os.path.exists(os.path.join(ROOT, r, name))), None)
    if not path:
        return None
    try:
        with open(os.path.join(ROOT, path)) as fh:
            s = json.load(fh)
        out = {}
        for k, p in s["pmc"].items():
            v = s.get("hbm_traffic", {}).get(k, {})
            e = {"traffic": int(v.get("hbm_read_bytes_per_launch", 0) + v.get("hbm_write_bytes_per_launch", 0))}
            get = lambda c: p[c]["full_batch_mean"] if c in p else None
            if get("TCC_HIT_sum") is not None and get("TCC_MISS_sum") is not None:
                e["l2_requests"] = int(get("TCC_HIT_sum") + get("TCC_MISS_sum"))
                e["l2_misses"] = int(get("TCC_MISS_sum"))
            if get("SQ_WAIT_ANY") is not None and get("SQ_WAVE_CYCLES"):
                e["wait_frac"] = round(get("SQ_WAIT_ANY") / get("SQ_WAVE_CYCLES"), 3)
            if get("SQ_ACTIVE_INST_ANY") is not None and get("SQ_WAVE_CYCLES"):
                e["issue_frac"] = round(get("SQ_ACTIVE_INST_ANY") / get("SQ_WAVE_CYCLES"), 3)
            if get("SQ_INSTS_VALU") is not None:
                e["valu_insts"] = get("SQ_INSTS_VALU")
                if "total" in p["SQ_INSTS_VALU"] and s.get("batches_mapped_by_the_pmc_runs"):  # all launches of the kernel in a step (the DP lists: several)
                    e["valu_insts_per_step"] = p["SQ_INSTS_VALU"]["total"] / s["batches_mapped_by_the_pmc_runs"]
            if get("SQ_LDS_BANK_CONFLICT") is not None and get("SQ_LDS_IDX_ACTIVE"):
                e["lds_conflict_frac"] = round(get("SQ_LDS_BANK_CONFLICT") / get("SQ_LDS_IDX_ACTIVE"), 4)
            if s.get("batches_mapped_by_the_pmc_runs") and "FETCH_SIZE" in p and "WRITE_SIZE" in p and "total" in p["FETCH_SIZE"] and "total" in p["WRITE_SIZE"]:
                # every launch of the kernel in a step (FETCH_SIZE / WRITE_SIZE are KiB; their two passes map the same batches)
                e["hbm_bytes_per_step"] = (p["FETCH_SIZE"]["total"] + p["WRITE_SIZE"]["total"]) * 1024.0 / s["batches_mapped_by_the_pmc_runs"]
                e["hbm_written_per_step"] = p["WRITE_SIZE"]["total"] * 1024.0 / s["batches_mapped_by_the_pmc_runs"]
            tr = s.get("kernel_trace", {}).get(k)
            if tr:
                e["trace_ms"] = round(tr["full_batch_avg_us"] / 1e3, 4)  # a full-batch launch of the kernel in the committed kernel trace
            nb = s.get("batches_mapped_by_the_pmc_runs")
            any_counter = next(iter(p.values()), None)
            if nb and any_counter:
                e["launches_per_step"] = round(any_counter["launches"] / nb, 2)
            out[k] = e
        sha = s.get("kernel_source_sha")
        return {"kernels": out, "file": path, "source_sha": sha, "stale": bool(sha) and sha != kernel_source_sha()}
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None


def essential_bytes(kernel, d, args):
    """Bytes per launch a kernel cannot avoid moving, from the run's own counters: what `frac` falls back
    on when no PMC pass of this workload is committed, and the yardstick beside the measured traffic when one is."""
    reads, steps = max(d["reads"], 1), max(args.steps, 1)
    h = d["sa_hits"] / reads                                # seed hits per read
    packed = 4.0 * ((args.rlen + 15) // 16 + (args.rlen + 31) // 32 + 2)  # 2-bit words + N masks of a read
    per_read = {
        "k_pack_reads": args.rlen + packed,
        # packed read in; per extension step one or two 16-byte rank records (fm_blocks counts them); per hit a suffix-array
        # entry and the hit record out; the direct comparison reads the 2-bit genome under the seed (E bases / 4)
        "k_seed": packed + 16.0 * d["fm_blocks"] / reads + (8 + 16) * h + d["fm_ext_steps"] / reads / 4.0,
        "k_cluster": 16.0 * h + 32.0,                       # hits in, a candidate out
        "k_build": 16.0 * h + 32.0 + 16.0 * (2 * h + 1) + 2 * args.rlen / 4.0,   # hits + candidate in, fragments out, gap bases compared
        "k_finish": 32.0 + 16.0 * (2 * h + 1) + 2 * args.rlen / 4.0 + 64.0 + 8.0,  # candidate + fragments in, columns compared, record + CIGAR out
    }.get(kernel, 0.0)
    return per_read * reads / steps


def dp_roofline(args, d, prof):
    """The DP stage against the chip's vector-instruction issue rate: `achieved` = cells x the sweep's instructions per cell
    over the stage's time (the lists' kernels share the chip on three streams: the stage is the sum of their work), `traffic`
    = the lane-operation slots the stage's kernels actually issued (SQ_INSTS_VALU x 64 summed over every DP launch of a step — the large tier's and the
    replay's included —, committed PMC pass) — idle lanes of
    ragged groups, staging and the tracebacks are the difference."""
    steps = max(args.steps, 1)
    ms = d["ms_dp"] / steps
    cells = d["dp_cells"] / steps
    x1 = bool(os.environ.get("MCX_DP_X1"))
    ops = (DP_OPS_PER_CELL_X1 if x1 else DP_OPS_PER_CELL)[args.alg]
    achieved = cells * ops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    kern = prof["kernels"] if prof else {}
    dpk = {k: v for k, v in kern.items() if k.startswith("k_dp_")}
    issued = sum(v.get("valu_insts_per_step", v.get("valu_insts", 0.0)) for v in dpk.values()) * 64 if dpk else None
    r = {"bound": "valu", "kernel": ("k_dp_lane<K, alg> (one problem per lane" if x1 else "k_dp_lane2<K, alg> (two problems per lane in 16-bit halves") + "; five lists on three streams, k_dp_sel<16> for targets above 256 bases)",
         "achieved": round(achieved, 2), "peak": round(VALU_PEAK_TLANEOPS, 1), "unit": "T lane-ops/s", "frac": round(achieved / VALU_PEAK_TLANEOPS, 4),
         "traffic": None if not issued else round(issued),
         "traffic_unit": None if not issued else f"vector lane-operation slots issued per step by the DP kernels (rocprofv3 SQ_INSTS_VALU x 64, {prof['file']})",
         "avg_launch_ms": round(ms, 3), "cells_per_step": round(cells), "ops_per_cell": ops, "ops_per_cell_is": "the vector instructions of the compiled sweep's inner loop per cell (this kernel's own count)",
         "ops_per_cell_minimal": DP_OPS_PER_CELL_MINIMAL[args.alg],
         "frac_at_minimal_ops": round(cells * DP_OPS_PER_CELL_MINIMAL[args.alg] / (ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS, 4) if ms > 0 else 0.0,
         "gcups": round(cells / max(ms, 1e-9) / 1e6, 1),
         "basis": "algorithmic lane-operations (cells of all problems x the sweep's vector instructions per cell) over the DP stage's live time, against "
                  "256 CU x 4 SIMD-32 x 2.4 GHz (a wave64 instruction per 2 cycles)"}
    if dpk:
        r["per_kernel"] = {k: {kk: v[kk] for kk in ("issue_frac", "wait_frac", "lds_conflict_frac", "valu_insts") if kk in v} for k, v in dpk.items()}
        if issued:
            r["issued_frac_of_peak"] = round(issued / (ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS, 4)
    # Two problems per lane halved the stage's instructions, and what it moves became the nearer wall: the lanes' words — traceback bits, strip edges, query codes, a
    # few GB live at once, far more than L2 holds — go out to HBM and come back.  When the committed PMC pass has the stage's bytes and they are the larger fraction of
    # their peak, the line's roofline is the memory one and the instruction view stays beside it.
    hbm = sum(v.get("hbm_bytes_per_step", 0.0) for v in dpk.values()) if dpk else 0.0
    if hbm and ms > 0:
        gbs = hbm / (ms * 1e-3) / 1e9
        valu = {k: r[k] for k in ("achieved", "peak", "unit", "frac", "traffic", "traffic_unit", "ops_per_cell", "ops_per_cell_minimal", "frac_at_minimal_ops", "issued_frac_of_peak") if k in r}
        r["valu_issue"] = valu
        if gbs / HBM_PEAK_GBS > r["frac"]:
            algorithmic = cells * (0.25 if args.alg == "nw" else 0.5) * 2 + cells * 0.25 * 2  # traceback bits written and read once; strip edges (two 16-bit values per row and 16 columns) written and read
            r.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": round(hbm),
                      "traffic_unit": f"HBM bytes per step of all DP kernels (rocprofv3 FETCH_SIZE + WRITE_SIZE over every launch, {prof['file']}); "
                                      f"{round(sum(v.get('hbm_written_per_step', 0.0) for v in dpk.values()) / 1e9, 1)} GB of them written",
                      "algorithmic_bytes_per_launch": round(algorithmic), "algorithmic_frac": round(algorithmic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "basis": "measured HBM bytes of the stage's kernels (committed PMC pass of this command) over the stage's live time; `algorithmic`: traceback bits and strip "
                               "edges of the problems' own cells, written and read once — the rest is cells of padding (rows and strips of a wave's longest problem, 16-column strips) and "
                               "what does not stay in L2 between its store and its fetch"})
    return r


def roofline(args, d, reads_per_s):
    """The dominant KERNEL's roofline.  A per-pair / seeding kernel: `frac` = measured HBM bytes of the launch / its live duration /
    the HBM peak (at most 1 by construction).  The DP stage (the longest one of BASELINE config 5): vector-instruction issue
    (dp_roofline).  A step that is a string of short launches (config 2) is bound by none of its kernels: `launch_bound` says so.
    The reference's own walk priced at our launch time (SURVEY 8d's figure, which can exceed 1 because the kernel does not do that
    walk) is kept as `speed_of_light_equiv`; `per_kernel` and `request_rate` list every kernel of a step by name — live time where
    the library times the kernel by itself, the committed kernel trace's otherwise."""
    steps = max(args.steps, 1)
    prof = pmc_profile(args)
    stale = bool(prof and prof.get("stale"))
    if stale:  # counters of other kernel sources: not this run's bytes — said, and not used
        print(f"bench.py: {prof['file']} was collected from kernel sources {prof['source_sha']}, this tree is {kernel_source_sha()}: its traffic is NOT used "
              "(re-run scripts/collect_profile.sh)", file=sys.stderr)
        stale_file, prof = prof["file"], None
    kern = prof["kernels"] if prof else {}
    nw = "true" if args.alg == "nw" else "false"
    stage_of, live = {}, {}
    for st, ks in STAGE_KERNELS.items():
        for k in ks:
            stage_of[k.format(nw=nw)] = st
        if len(ks) == 1 and d.get(st, 0) > 0:
            live[ks[0]] = d[st] / steps  # (k_build: its two launches together)
    longest = max(live, key=live.get)
    reads_step = d["reads"] / steps
    seed_bytes = 64.0 * 1.107 * d["fm_ext_steps"] + float(d["reads"]) * args.rlen
    seed_ms = d["ms_seed"] / steps
    step_ms = sum(d[k] for k in d if k.startswith("ms_") and k not in ("ms_total",)) / steps
    dp_ms = d.get("ms_dp", 0) / steps
    launch_bound = step_ms < 4.0 and max(live[longest], dp_ms) < 0.4 * step_ms  # a string of short launches: no kernel's (or stage's) roofline describes the step
    if launch_bound:
        r = {"bound": "launch", "kernel": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
             "basis": "the step is a string of launches of a few tens to hundreds of microseconds each on a chip that this batch does not fill: it is bound by launch "
                      "and tail latency, not by any kernel's traffic or arithmetic (see launch_bound)"}
    elif dp_ms > live[longest]:
        r = dp_roofline(args, d, prof)
    else:
        traffic = kern.get(longest, {}).get("traffic")
        ess = essential_bytes(longest, d, args)
        moved = traffic if traffic else ess
        achieved = moved / (live[longest] * 1e-3) / 1e9
        r = {"bound": "hbm", "kernel": longest, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
             "traffic_unit": None if not traffic else f"HBM bytes per launch (rocprofv3 FETCH_SIZE + WRITE_SIZE, {prof['file']})",
             "traffic_from": None if not traffic else f"{prof['file']} (committed rocprofv3 --pmc pass of this command, kernel sources {prof.get('source_sha') or 'not recorded'}); the duration is live",
             "avg_launch_ms": round(live[longest], 3),
             "basis": "measured HBM bytes of the launch (committed PMC pass of this command) over the live launch time" if traffic else
                      "no PMC pass of this workload is committed: the kernel's essential bytes (DESIGN.md §3) over the live launch time",
             "algorithmic_bytes_per_launch": round(ess), "algorithmic_frac": round(ess / (live[longest] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    # every kernel of a step, by name
    per, req, launches = {}, {}, 0.0
    names = sorted(set(list(live) + [k for k in kern if k.startswith(STEP_KERNEL_PREFIXES)]))
    for k in names:
        p = kern.get(k, {})
        ms = live.get(k) if k in live and k != "k_build" else p.get("trace_ms")  # per launch
        launches += p.get("launches_per_step", 0.0)
        e = {"stage": (stage_of.get(k) or ("ms_dp" if k.startswith("k_dp_") else None) or "")[3:] or None,
             "ms_live": None if k not in live else round(live[k], 3), "ms_per_launch_trace": p.get("trace_ms"), "launches_per_step": p.get("launches_per_step")}
        if p.get("traffic") and ms:
            e.update({"hbm_bytes_per_launch": p["traffic"], "hbm_gbs": round(p["traffic"] / (ms * 1e-3) / 1e9, 1),
                      "frac_of_peak": round(p["traffic"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3), "bytes_per_read": round(p["traffic"] / reads_step, 1)})
        eb = essential_bytes(k, d, args)
        if eb:
            e["essential_bytes_per_read"] = round(eb / reads_step, 1)
        for c in ("issue_frac", "wait_frac", "lds_conflict_frac"):
            if c in p:
                e[c] = p[c]
        per[k] = e
        if "l2_requests" in p and ms:
            q = {"l2_requests_per_launch": p["l2_requests"], "g_per_s": round(p["l2_requests"] / (ms * 1e-3) / 1e9, 1), "ceiling_g_per_s": GATHER_CEILING_G_PER_S,
                 "frac": round(p["l2_requests"] / (ms * 1e-3) / 1e9 / GATHER_CEILING_G_PER_S, 3), "requests_per_read": round(p["l2_requests"] / reads_step, 2)}
            if "l2_misses" in p:
                q.update({"l2_misses_per_launch": p["l2_misses"], "misses_per_read": round(p["l2_misses"] / reads_step, 2),
                          "misses_g_per_s": round(p["l2_misses"] / (ms * 1e-3) / 1e9, 1), "walk_miss_ceiling_g_per_s": WALK_MISS_CEILING_G_PER_S,
                          "miss_frac_of_ceiling": round(p["l2_misses"] / (ms * 1e-3) / 1e9 / WALK_MISS_CEILING_G_PER_S, 3)})
            req[k] = q
    if stale:
        r["traffic_stale"] = f"{stale_file} predates the kernel sources of this build"
    r.update({"per_kernel": per, "request_rate": req,
              "launches_per_step": round(launches, 1) if launches else None,
              "speed_of_light_equiv": {"kernel": "k_seed", "bytes_per_read": round(seed_bytes / max(d["reads"], 1), 1),
                                       "gbs": round(seed_bytes / steps / (seed_ms * 1e-3) / 1e9, 1) if seed_ms > 0 else None,
                                       "ratio_to_peak": round(seed_bytes / steps / (seed_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3) if seed_ms > 0 else None,
                                       "note": "SURVEY 8d's seeding bytes (64*1.107*E + rlen per read: the reference's FM walk) over our launch time; not a utilisation — "
                                               "the kernel reaches the same seeds through a K-mer jump table and direct genome comparison"},
              "path": path_roofline(d, args, reads_per_s)})
    # a step of many short launches: no kernel's roofline describes it
    if launch_bound:
        r["launch_bound"] = {"ms_per_step": round(step_ms, 3), "longest_kernel": longest, "longest_kernel_ms": round(live[longest], 3), "dp_stage_ms": round(dp_ms, 3),
                             "launches_per_step": round(launches, 1) if launches else None,
                             "mean_us_per_launch": round(1e3 * step_ms / launches, 1) if launches else None,
                             "note": "the step is a string of launches of a few tens to hundreds of microseconds each on a chip that this batch does not fill: "
                                     "it is bound by launch and tail latency, not by any kernel's traffic — `frac` of the longest kernel says nothing here"}
    return r


def path_roofline(d, args, reads_per_s):
    reads = max(d["reads"], 1)
    e, h = d["fm_ext_steps"] / reads, d["sa_hits"] / reads
    b_read = 64.0 * (1.107 * e + 31.0 * h) + 8.0 * h + args.rlen + args.rlen / 4.0
    gbs = reads_per_s * b_read / 1e9
    return {"bytes_per_read": round(b_read, 1), "equiv_gbs": round(gbs, 1), "ratio_to_peak": round(gbs / HBM_PEAK_GBS, 4),
            "bound_reads_per_s": round(HBM_PEAK_GBS * 1e9 / b_read, 1),
            "note": "SURVEY 8d's whole-path bytes of the reference's algorithm (B_read) at our rate: a speed-of-light comparison, not a utilisation"}


def pcie_inclusive(args, mapper, batches, reads_per_step, dist, dev):
    """The same steps with the device boundary the drop-in has: reads in pinned host memory in — packed to 2 bits the way the
    file front end's parser hands them over (mcx_stream_submit_packed) —, records in pinned host memory out, the copies of one
    batch under the kernels of its neighbours (Mapper.map_stream_packed)."""
    from mapcaller_amd import api
    k = args.pcie_steps
    host = []
    for b in batches[:min(k, len(batches), 6)]:  # (six distinct batches in pinned host memory, in turn)
        words, lens, odd, n_odd, row_words = api.pack_reads(b.reshape(reads_per_step, args.rlen))
        host.append((words, lens, odd, n_odd, row_words))
    nd = getattr(args, "native_dir", None)
    if nd:  # the same batches for the process on the system's runtime (native_boundary below)
        try:
            meta = json.load(open(os.path.join(nd, "meta.json")))
            meta["batches"] = [{"row_words": rw, "n_odd": n} for (_, _, _, n, rw) in host]
            for i, (w, l, o, _, _) in enumerate(host):
                w.numpy().tofile(os.path.join(nd, f"batch{i}.words")); l.numpy().tofile(os.path.join(nd, f"batch{i}.lens")); o.numpy().tofile(os.path.join(nd, f"batch{i}.odd"))
            json.dump(meta, open(os.path.join(nd, "meta.json"), "w"))
        except OSError:
            args.native_dir = None
    packed = [(w.data_ptr(), rw, l.data_ptr(), o.data_ptr(), n) for (w, l, o, n, rw) in host]
    packed = [packed[i % len(packed)] for i in range(k)]  # (more steps than resident batches: the batches come round again)
    outs = mapper.stream_outputs(reads_per_step, 3, 32)  # (page-locking gigabytes takes seconds: not part of the path)
    b0 = mapper.map_stream_packed(packed[:3], reads_per_step, True, outs, out32=True)  # the three slots in HBM, streams, events: made on first use
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b1 = mapper.map_stream_packed(packed, reads_per_step, True, outs, out32=True)
    n_bytes = (b1[0] - b0[0], b1[1] - b0[1])
    dt = time.perf_counter() - t0
    if dist:
        from mapcaller_amd import dist as mdist_
        dt = mdist_.max_over_ranks(dt, dev)
    world = dist.get_world_size() if dist else 1
    return {"value": round(k * reads_per_step * world / dt, 1), "unit": "reads/s", "steps": k, "ms_per_step": round(1000 * dt / k, 3),
            "h2d_bytes_per_read": round(n_bytes[0] / (k * reads_per_step), 1), "d2h_bytes_per_read": round(n_bytes[1] / (k * reads_per_step), 1),
            "note": "2-bit reads + lengths from pinned host memory (the form the file front end's parser hands over; the ASCII bytes are restored "
                    "exactly on the device), alignment records (32 bytes each: mcx_aln32, packed on the device — at 64 the copy out outlasted the stretch of the next step in which "
                    "the host waits for nothing, and the step took 20.5 instead of 17.3 ms) + CIGAR pool back to pinned host memory; copies of batch i+1 / i-1 overlap the kernels "
                    "of batch i on separate HIP streams (mcx_stream_*); the first copy in and the last copy out of the sequence have nothing to "
                    "hide behind and are part of the time (about 18 ms per sequence at 8 M reads a step)"}


def native_boundary(args, in_process):
    """value_pcie_inclusive's steps from a process that runs on the system's HIP runtime (python -m mapcaller_amd.boundary: libmcx.so through
    ctypes, no torch) — the situation of a C/C++ host: the CLI, the reference with INTEGRATION.md's binding.  This process has let go of the GPU's memory."""
    r = subprocess.run([sys.executable, "-m", "mapcaller_amd.boundary", args.native_dir], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.abspath(__file__)))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not lines:
        return {"error": (r.stderr or r.stdout)[-300:]}
    o = json.loads(lines[-1])
    o["note"] = ("the same batches, steps and buffers as value_pcie_inclusive from a process of its own on the system's HIP runtime (%s; this process runs on the one torch's wheel carries, "
                 "%s, which puts the copies of both directions on one SDMA engine: 320 MB in + 256 MB out take 10.6 ms together there, 5.96 ms on the system's — "
                 "scripts/probe/d2h_probe.*); in-process figure: %s ms per step" % (o.get("hip_runtime_version"), torch.version.hip, in_process.get("ms_per_step") if in_process else None))
    return o


def file_to_file(args, index, kept, reads_per_step):
    """FASTQ files in, SAM file out through mcx_map_files_ex — what the CLI runs once its index is loaded —, both in tmpfs: the
    reads of --file-batches batches of the timed region as two FASTQ files (16 M reads by default, eight batches of the pipeline: the filling
    and draining of parse | map | format | write is a small part of the run), mapped in batches of --file-batch-reads."""
    import shutil
    from mapcaller_amd import api, synth
    root = "/dev/shm" if os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(prefix="mcx_f2f_", dir=root)
    try:
        f1, f2, sam = os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq"), os.path.join(tmp, "o.sam")
        for k, batch in enumerate(kept):
            reads = batch.reshape(reads_per_step, args.rlen).cpu()
            synth.write_fastq(f1, reads, 0, 2, prefix=f"b{k}", append=k > 0)
            synth.write_fastq(f2, reads, 1, 2, prefix=f"b{k}", append=k > 0)
            del reads
        mp = api.Mapper(index, alg=args.alg, max_read_len=max(256, args.rlen), max_batch_reads=args.file_batch_reads)
        runs = []
        for _ in range(2):  # (the first run also page-locks the batch buffers)
            mp.reset()
            if os.path.exists(sam):
                os.remove(sam)  # (giving 6 GB of tmpfs pages back is the file system's business, not the run's)
            t0 = time.perf_counter()
            st = mp.map_files(f1, f2, sam, threads=args.file_threads)
            runs.append(time.perf_counter() - t0)
        mp.reset()
        t0 = time.perf_counter()
        mp.map_files(f1, f2, None, threads=args.file_threads)  # the same without the SAM file: what the input side and the device sustain
        t_in = time.perf_counter() - t0
        out = {"value": round(st["reads"] / runs[1], 1), "unit": "reads/s", "reads": st["reads"], "seconds": round(runs[1], 4), "first_run_seconds": round(runs[0], 4),
               "without_sam_output": {"value": round(st["reads"] / t_in, 1), "seconds": round(t_in, 4),
                                      "note": "FASTQ files in, records left in host memory: parse + pack + copies + mapping; the difference is SAM text and its way into ONE "
                                              "file (writers of one file queue behind its lock: one writer of a growing tmpfs file gets 5.7-6.6 GB/s on this box, 3 GB of text per 8 M reads)"},
               "device_stage_ms_per_batch": {k[3:]: round(st[k] / max(1, -(-st["reads"] // args.file_batch_reads)), 3) for k in st if k.startswith("ms_")},
               "fastq_bytes": os.path.getsize(f1) + os.path.getsize(f2), "sam_bytes": os.path.getsize(sam), "batch_reads": args.file_batch_reads,
               "host_threads": args.file_threads or "default (three quarters of the CPUs the process is given per pool: %d of usable_cpus() = %d)" % (max(1, usable_cpus() * 3 // 4), usable_cpus()), "where": tmp.rsplit("/", 1)[0],
               "batches_of_the_timed_region": len(kept),
               "note": "two plain FASTQ files -> one SAM file, index already in HBM (the CLI loads it once per run); parse + 2-bit packing, "
                       "copies, mapping, SAM text and positioned writes overlapped (mcx_files.cpp)"}
        mp.close()
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def vcf_leg(args, index, mapper, batches, off, d_aln, d_cig, reads_per_step, n_steps, d, dist, dev, rank, world):
    from mapcaller_amd import api, dist as mdist
    G = index.genome_size
    # this leg keeps the ten planes of the genome (68 GB at 3.1 Gbp; 124 GB until round 5) and per-read alignment detail in HBM:
    # the timed region's context and all but one batch make room
    first, timed = batches[0], (batches[1:] or batches[:1])  # (one batch only: it is mapped again, and the duplicate cap then refuses most of its reads)
    if mapper is not None:
        mapper.close()
    torch.cuda.empty_cache()  # (the caching allocator would sit on the freed batches)
    slice_reads = min(reads_per_step, args.vcf_slice_reads)
    # (the pair records stay — round 4 gave them back here, and mapped in slices of 4 M reads —; the large tier's records sized for this workload's
    #  heavy pairs, 0.7 % of a batch, instead of config 5's 4.5 %: 8 GB instead of 24)
    if slice_reads * 3072 > args.vcf_tier1_gb << 30:  # (the library's own size for the tier: 3 KB per read of the batch, 2 - 24 GB)
        os.environ["MCX_TIER1_GB"] = str(args.vcf_tier1_gb)
    try:
        mapper = api.Mapper(index, alg=args.alg, max_read_len=max(256, args.rlen), max_batch_reads=slice_reads)
    finally:
        os.environ.pop("MCX_TIER1_GB", None)
    planes = api.planes_alloc(G, dev)

    def map_batch(b):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for lo in range(0, reads_per_step, slice_reads):
            n = min(slice_reads, reads_per_step - lo)
            mapper.map_batch_dev(b.data_ptr() + lo * args.rlen, off.data_ptr(), n, True, d_aln.data_ptr(), d_cig.data_ptr())
        torch.cuda.synchronize()
        return time.perf_counter() - t

    # a run without the bookkeeping and one with it, each from its first batch (the insert-size estimate from its first pairs, their replay);
    # the batches behind the first are timed: a batch of the run in its steady state, like a step of the timed region above
    map_batch(first)        # (first use of the context: allocations)
    mapper.reset()
    map_batch(first)
    t_plain = sum(map_batch(b) for b in timed) / len(timed)
    mapper.reset()
    mapper.profile_attach(planes.data_ptr())
    map_batch(first)        # (first use of the bookkeeping: its allocations, the first growth of the record archive)
    planes.zero_()
    mapper.reset()
    mapper.profile_attach(planes.data_ptr())
    t_first = map_batch(first)
    # (a batch's bookkeeping is queued behind it and runs under the next batch's kernels: the batches follow one another without a wait in between; the time
    #  of a batch in the run's steady state is the time between two calls' returns; what is left of the last batch's bookkeeping when its call returns is the tail)
    torch.cuda.synchronize()
    marks = [time.perf_counter()]
    for b in timed:
        for lo in range(0, reads_per_step, slice_reads):
            n = min(slice_reads, reads_per_step - lo)
            mapper.map_batch_dev(b.data_ptr() + lo * args.rlen, off.data_ptr(), n, True, d_aln.data_ptr(), d_cig.data_ptr())
        marks.append(time.perf_counter())
    torch.cuda.synchronize()
    t_tail = time.perf_counter() - marks[-1]
    t_acc = (marks[-1] - marks[1]) / (len(marks) - 2) if len(marks) > 2 else marks[-1] - marks[0] + t_tail
    t_seq = (marks[-1] - marks[0] + t_tail) / len(timed)
    hbm_free, hbm_total = torch.cuda.mem_get_info(dev)  # (the leg is the fullest the device gets: index, planes, a full-batch context, the per-read detail)
    t_sp = time.perf_counter()
    sparse = mapper.profile_sparse_raw(shard=world > 1, copy=False)  # the tally records leave HBM here, once
    t_sp = time.perf_counter() - t_sp
    t_settle = time.perf_counter()
    mapper.profile_settle()  # once per run: the planes kept as differences become counts
    torch.cuda.synchronize()
    t_settle = time.perf_counter() - t_settle
    if dist:
        dist.barrier()
    t2 = time.perf_counter()
    # (every rank accumulated a run of its own here, so the readCount planes are summed too)
    planes, merged = mdist.reduce_profile(planes, sparse, G, root=0, shared_read_count=False, mapper=mapper)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t_red = time.perf_counter() - t2
    tot = mdist.sum_over_ranks([d["pairs"], d["pair_dist_sum"], d["pair_len_sum"]], dev)
    gb = (mdist.reduce_profile.last_bytes if world > 1 else api.planes_stride(G) * 22) / 1e9  # (one GPU: what a rank would put on the wire, the readCount plane included here)
    vcf = {"profile_batch_ms": round(1000 * t_acc, 2), "same_batches_without_profile_ms": round(1000 * t_plain, 2),
           "profile_overhead_ms": round(1000 * (t_acc - t_plain), 2), "batches_timed": len(timed), "sequence_ms_per_batch_tail_included": round(1000 * t_seq, 2), "last_batch_bookkeeping_tail_ms": round(1000 * t_tail, 2), "first_batch_of_the_run_ms": round(1000 * t_first, 2),
           "batches": 1 + len(timed), "hbm_free_gb": round(hbm_free / 1e9, 1), "hbm_total_gb": round(hbm_total / 1e9, 1), "slice_reads": slice_reads, "pair_records_kept": bool(args.full_sa >= 2), "tier1_gb": args.vcf_tier1_gb,
           "note": "per batch of the run in its steady state (the estimate carried from batch to batch, as in the timed region): mean over the batches behind "
                   "the first; planes and sparse records hold all of them", "sparse_records_to_host_ms": round(1000 * t_sp, 2),
           "settle_ms_once_per_run": round(1000 * t_settle, 2),
           "reduce_ms": round(1000 * t_red, 2), "reduce_gb": round(gb, 2),
           "reduce_gbs_into_root": None if world == 1 else round(gb * (world - 1) / max(t_red, 1e-9), 1),
           "reduce": "none (one GPU); 20-22 bytes per position and rank when there are several" if world == 1 else
                     f"RCCL reduce of {world} x {round(gb * 1e9 / G, 1)} bytes per position onto rank 0 in 1-GiB pieces (the planes as they lie in HBM: multi_hit in 32 bits, the other nine in 16)",
           "sparse_records": len(merged)}
    if rank == 0:  # VariantCalling() runs once, on the reduced profile
        mapper.profile_finalize(planes.data_ptr())
        cov = 0
        _, half = api.planes_parts(planes, G)
        for lo in range(0, G, 1 << 28):  # (in pieces: a genome-sized temporary does not fit beside the planes)
            hi = min(G, lo + (1 << 28))
            cov += int(((half[0, lo:hi] | half[1, lo:hi] | half[2, lo:hi] | half[3, lo:hi]) != 0).sum().item())
        vcf["covered_positions"] = cov
        with tempfile.TemporaryDirectory() as tmp:
            vs = index.call_variants(planes.data_ptr(), merged, tot[0], tot[1], tot[2], os.path.join(tmp, "bench.vcf"),
                                     ref_name="synthetic", cmdline="bench.py")
        # both scans stream the planes once: 8 B (k_vc_depth: A C G T in 16 bits each) and 12 B (those and multi_hit) + 2-bit base + depth word (k_vc_scan) per position
        vcf["call_variants"] = {"ms_total": round(vs["ms_total"], 2), "k_vc_depth_ms": round(vs["ms_depth"], 3),
                                "k_vc_scan_ms": round(vs["ms_scan"], 3), "records": vs["n_records"], "snv": vs["n_snv"],
                                "k_vc_depth_gbs": round(8.0 * G / max(vs["ms_depth"], 1e-6) / 1e6, 1),
                                "k_vc_scan_gbs": round(12.29 * G / max(vs["ms_scan"], 1e-6) / 1e6, 1)}
    mapper.close()
    return vcf


def other_genome(args):
    """Two steps against the other kind of synthetic genome, as a child process once this one has let go of the GPU's memory."""
    kind = "uniform" if args.genome == "human" else "human"
    cmd = [sys.executable, os.path.abspath(__file__), "--genome", kind, "--second-genome", "0", "--steps", "2", "--warmup", "1", "--cpu-pairs", "0",
           "--vcf-reduce", "0", "--other-configs", "0", "--file-steps", "0", "--pcie-steps", "0", "--full-line", "1", "--detail-stdout", "0", "--detail-tag", "other_genome", "--genome-mbp", str(args.genome_mbp), "--contigs", str(args.contigs), "--batch-pairs", str(args.batch_pairs),
           "--rlen", str(args.rlen), "--sub", str(args.sub), "--ins", str(args.ins), "--dele", str(args.dele), "--alg", args.alg, "--full-sa", str(args.full_sa)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    o = json.loads(line)
    return {"genome": kind, "value": o["value"], "unit": o["unit"], "steps": o["steps"], "ms_per_step": o["ms_per_step"], "workload": o["config"]["workload"],
            "per_read": o["per_read"], "stage_ms_per_step": o["stage_ms_per_step"], "tier1_pairs": o["tier1_pairs"],
            "value_pcie_inclusive": o.get("value_pcie_inclusive"),
            "roofline": {k: o["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "basis")}}


def other_configs(args):
    """BASELINE.json's configs 5 and 2, each as a child process once this one has let go of the GPU's memory: two timed steps, stage
    times, DP cell updates per second, and the CPU baseline on a small sample of the same reads."""
    runs = [
        ("config 5: GRCh38-sized genome, 250 bp PE at 5 % indels per base, -alg nw",
         ["--genome", args.genome, "--genome-mbp", str(args.genome_mbp), "--contigs", str(args.contigs), "--batch-pairs", str(args.batch_pairs), "--rlen", "250",
          "--sub", str(args.sub), "--ins", "0.025", "--dele", "0.025", "--alg", "nw", "--cpu-pairs", "300000", "--cpu-level", "sam"]),
        ("config 2: E. coli-sized genome (4.6 Mbp, one contig), 1 M x 100 bp SE, -alg ksw2",
         ["--genome", "uniform", "--genome-mbp", "4.6", "--contigs", "1", "--repeats", "20", "--batch-pairs", "1000000", "--single-end", "1", "--rlen", "100",
          "--sub", str(args.sub), "--ins", str(args.ins), "--dele", str(args.dele), "--alg", "ksw2", "--cpu-pairs", "20000000", "--cpu-level", "full"]),
    ]
    res = []
    for name, extra in runs:
        # (two warm-up batches: the first makes the large tier's records and — when it ends — the long DP list's scratch grow to the workload, mcx.h; the second runs on them)
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "2", "--second-genome", "0", "--other-configs", "0", "--vcf-reduce", "0",
               "--pcie-steps", "0", "--file-steps", "0", "--full-line", "1", "--detail-stdout", "0", "--detail-tag", "cfg" + name.split(":")[0].split()[-1], "--full-sa", str(args.full_sa)] + extra
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=900)
            o = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            res.append({"config": name, "value": o["value"], "unit": o["unit"], "steps": o["steps"], "ms_per_step": o["ms_per_step"], "workload": o["config"]["workload"],
                        "stage_ms_per_step": o["stage_ms_per_step"], "per_read": o["per_read"], "dp": o["dp"], "tier1_pairs": o["tier1_pairs"],
                        "halved_selections": o["halved_selections"], "cpu_baseline": o.get("cpu_baseline"),
                        "simple_pairs": o.get("simple_pairs"),
                        "roofline": {k: o["roofline"].get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_unit", "avg_launch_ms", "gcups", "cells_per_step",
                                                                       "ops_per_cell", "ops_per_cell_minimal", "frac_at_minimal_ops", "issued_frac_of_peak", "launch_bound", "launches_per_step", "basis", "valu_issue",
                                                                       "algorithmic_bytes_per_launch", "algorithmic_frac")}})
        except Exception as e:
            res.append({"config": name, "error": str(e)[:200]})
    return res


LINE_LIMIT = 6000  # bytes of the final stdout line: the driver keeps the last 8 KB of stdout, and a line it cannot see whole it cannot parse (round 5: 33.6 KB, `parsed: null`)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None} if isinstance(d, dict) else None


def compact_line(out, detail_path=None):
    """The ONE line the driver parses: the contract's keys, the dominant kernel's roofline and the CPU baseline as numbers, the other legs as
    their headline figures.  Everything else — per-kernel tables, request rates, the prose — is the detail record (`detail`: its path; also printed
    as an earlier stdout line).  Never longer than LINE_LIMIT: sections are dropped from the end of `order` until it fits."""
    cfg = out.get("config", {})
    wl = cfg.get("workload", "")
    wl = re.sub(r" \(\d+ contigs, .*?; GRCh38 itself is unavailable offline\)", " (synthetic: GRCh38 is unavailable offline; repeat landscape in the detail record)", wl)
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": wl[:400], **(_pick(cfg, ("reads_per_step_per_gpu", "index_hbm_gb", "multi_gpu_host_ms_per_step")) or {})}
    line["config"]["multi_gpu"] = None if not cfg.get("multi_gpu") else str(cfg["multi_gpu"])[:260]
    r = out.get("roofline") or {}
    line["roofline"] = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic")}
    line["roofline"].update(_pick(r, ("traffic_from", "traffic_stale", "avg_launch_ms", "algorithmic_bytes_per_launch", "algorithmic_frac", "gcups", "ops_per_cell", "frac_at_minimal_ops",
                                      "launches_per_step")) or {})
    if isinstance(r.get("valu_issue"), dict):
        line["roofline"]["valu_issue"] = _pick(r["valu_issue"], ("achieved", "peak", "unit", "frac", "ops_per_cell"))
    if isinstance(line["roofline"].get("kernel"), str):
        line["roofline"]["kernel"] = line["roofline"]["kernel"][:80]
    if r.get("launch_bound"):
        line["roofline"]["launch_bound"] = _pick(r["launch_bound"], ("ms_per_step", "longest_kernel", "longest_kernel_ms", "launches_per_step", "mean_us_per_launch"))
    c = out.get("cpu_baseline")
    if isinstance(c, dict):
        line["cpu_baseline"] = _pick(c, ("value", "unit", "cores", "kind", "error")) or {}
        if "sample" in c:
            line["cpu_baseline"]["sample"] = str(c["sample"])[:200]
        for sub in ("mapping_only", "single_thread"):
            if isinstance(c.get(sub), dict):
                line["cpu_baseline"][sub] = _pick(c[sub], ("value", "cores"))
    line["stage_ms_per_step"] = out.get("stage_ms_per_step")
    line["per_read"] = out.get("per_read")
    line.update(_pick(out, ("simple_pairs", "tier1_pairs", "replayed_pairs", "halved_selections")) or {})
    optional = {}
    p = out.get("value_pcie_inclusive")
    if isinstance(p, dict):
        optional["value_pcie_inclusive"] = _pick(p, ("value", "unit", "steps", "ms_per_step", "h2d_bytes_per_read", "d2h_bytes_per_read", "error")) or {}
        if isinstance(p.get("system_runtime"), dict):
            optional["value_pcie_inclusive"]["system_runtime"] = _pick(p["system_runtime"], ("value", "ms_per_step", "error"))
    f = out.get("value_file_to_file")
    if isinstance(f, dict):
        optional["value_file_to_file"] = _pick(f, ("value", "unit", "reads", "seconds", "error")) or {}
        if isinstance(f.get("without_sam_output"), dict):
            optional["value_file_to_file"]["without_sam_output"] = f["without_sam_output"].get("value")
    v = out.get("vcf_reduce")
    if isinstance(v, dict):
        optional["vcf_reduce"] = _pick(v, ("profile_batch_ms", "same_batches_without_profile_ms", "hbm_free_gb", "reduce_ms", "reduce_gb", "reduce_gbs_into_root", "sparse_records", "covered_positions", "error")) or {}
        if isinstance(v.get("call_variants"), dict):
            optional["vcf_reduce"]["call_variants_ms"] = v["call_variants"].get("ms_total")
            optional["vcf_reduce"]["call_variants_records"] = v["call_variants"].get("records")
    g = out.get("other_genome")
    if isinstance(g, dict):
        optional["other_genome"] = _pick(g, ("genome", "value", "ms_per_step", "error"))
    oc = out.get("other_configs")
    if isinstance(oc, list):
        optional["other_configs"] = []
        for e in oc:
            x = {"config": str(e.get("config", ""))[:90]}
            x.update(_pick(e, ("value", "unit", "ms_per_step", "error")) or {})
            if isinstance(e.get("roofline"), dict):
                x["roofline"] = _pick(e["roofline"], ("bound", "frac", "achieved", "peak", "unit", "gcups", "frac_at_minimal_ops", "algorithmic_frac"))
                if isinstance(e["roofline"].get("valu_issue"), dict):
                    x["roofline"]["valu_issue_frac"] = e["roofline"]["valu_issue"].get("frac")
            if isinstance(e.get("cpu_baseline"), dict):
                x["cpu_baseline"] = _pick(e["cpu_baseline"], ("value", "cores", "kind"))
            if isinstance(e.get("stage_ms_per_step"), dict):
                x["stage_ms_per_step"] = e["stage_ms_per_step"]
            optional["other_configs"].append(x)
    if detail_path:
        line["detail"] = detail_path
    order = ["value_pcie_inclusive", "value_file_to_file", "vcf_reduce", "other_configs", "other_genome"]
    for k in order:
        if k in optional:
            line[k] = optional[k]
    for k in reversed(order + ["per_read", "stage_ms_per_step"]):  # (cannot happen with the sections as they are; a guard, not a plan)
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(k, None)
    return line


def emit(out, args):
    """Detail record first (a file under --detail-dir; with --detail-stdout 1 also an earlier stdout line that does not start with `{`), then the compact line LAST."""
    path = None
    try:
        os.makedirs(args.detail_dir, exist_ok=True)
        path = os.path.join(args.detail_dir, "bench_detail.json" if not args.detail_tag else f"bench_detail_{args.detail_tag}.json")
        with open(path, "w") as fh:
            json.dump(out, fh)
        path = os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT) else path
    except OSError:
        path = None
    if args.detail_stdout:
        print("BENCH_DETAIL " + json.dumps(out), flush=True)
    line = json.dumps(out if args.full_line else compact_line(out, path))
    print(line, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome-mbp", type=float, default=3100.0, help="synthetic genome size (Mbp)")
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--batch-pairs", type=int, default=4_000_000, help="read pairs per step and per GPU")
    ap.add_argument("--rlen", type=int, default=150)
    ap.add_argument("--sub", type=float, default=0.005, help="per-base substitution rate of the simulated reads")
    ap.add_argument("--ins", type=float, default=0.001, help="per-base insertion rate")
    ap.add_argument("--dele", type=float, default=0.001, help="per-base deletion rate")
    ap.add_argument("--alg", default="ksw2", choices=["nw", "ksw2"])
    ap.add_argument("--full-sa", type=int, default=2, help="1: every suffix-array entry, the jump table and the rank records in HBM; 2: the pair records too (two bases per step of the seeding walk)")
    ap.add_argument("--cpu-pairs", type=int, default=-1,
                    help="pairs of the CPU-baseline sample (0 = skip, -1 = about 20 s of work for this host's core count)")
    ap.add_argument("--cpu-t1-pairs", type=int, default=-1, help="pairs of the -t 1 run of the CPU baseline (-1: about 20 s of its clock)")
    ap.add_argument("--cpu-level", default="full", choices=["sam", "two", "full"],
                    help="runs of the CPU baseline: with -sam only | and without any output (mapping_only) | and at -t 1")
    ap.add_argument("--repeats", type=int, default=2000, help="--genome uniform: planted dispersed repeat families")
    ap.add_argument("--genome", default="human", choices=["human", "uniform"],
                    help="repeat content of the synthetic genome: a human-like landscape (default) or round 1's nearly repeat-free one")
    ap.add_argument("--second-genome", type=int, default=1,
                    help="1: after the main run, map 2 steps against the other kind of genome as well and report them under `other_genome`")
    ap.add_argument("--native-boundary", type=int, default=1, help="the host-buffer leg once more from a process on the system's HIP runtime (mapcaller_amd/boundary.py: no torch in it), "
                                                                   "after this one has let go of the GPU's memory; N = 1 only; 0 = skip")
    ap.add_argument("--pcie-steps", type=int, default=36, help="steps of the host-buffer leg (value_pcie_inclusive), over the timed region's batches in turn; 0 = skip")
    ap.add_argument("--single-end", type=int, default=0, help="1: single-end reads (--batch-pairs then counts reads)")
    ap.add_argument("--other-configs", type=int, default=1,
                    help="1: after the main run, BASELINE.json's configs 5 (250 bp PE at 5 %% indels, -alg nw) and 2 (E. coli-sized genome, 1 M x 100 bp SE) "
                         "as child processes, reported under `other_configs` with their stage times, DP GCUPS and CPU baselines")
    ap.add_argument("--file-steps", type=int, default=1, help="1: the file-to-file leg (value_file_to_file): one batch as FASTQ files in tmpfs -> SAM; 0 = skip")
    ap.add_argument("--file-batches", type=int, default=2, help="batches of the timed region whose reads the file-to-file leg maps (2 x 8 M reads: eight batches of the file pipeline)")
    ap.add_argument("--file-batch-reads", type=int, default=1 << 21, help="reads per batch of the file front end's pipeline")
    ap.add_argument("--file-threads", type=int, default=0, help="host threads per pool of the file front end (0 = pick)")
    ap.add_argument("--vcf-slice-reads", type=int, default=8_000_000, help="reads per mapping call in the -vcf leg")
    ap.add_argument("--vcf-tier1-gb", type=int, default=8, help="HBM for the large tier's pair records in the -vcf leg (MCX_TIER1_GB)")
    ap.add_argument("--vcf-reduce", type=int, default=1,
                    help="after the timed region: accumulate the -vcf alignment profile of one batch and sum it over the "
                         "ranks with RCCL (1 = yes, 0 = no, -1 = only when more than one GPU)")
    ap.add_argument("--detail-dir", default=os.path.join(ROOT, "gpurun_out"), help="where the detail record (every section in full) is written; the final line names it")
    ap.add_argument("--detail-tag", default="", help="suffix of the detail record's file name")
    ap.add_argument("--detail-stdout", type=int, default=0, help="1: the detail record also as an earlier stdout line (`BENCH_DETAIL {...}`); off by default — nothing but the final line goes to stdout")
    ap.add_argument("--full-line", type=int, default=0, help="1: the final line is the detail record itself (what the child processes of this script read)")
    return ap.parse_args()


def _put_last(codes, idx, val, stamp):
    """codes[idx] = val where copies overlap one another: the LATER element of the call wins, whatever order the scatter's
    writes land in (plain index assignment resolves duplicate indices in no fixed order, which made the genome differ from
    process to process).  stamp: an int32 scratch tensor as long as codes."""
    e = torch.arange(idx.numel(), device=idx.device, dtype=torch.int32)
    stamp.index_fill_(0, idx, -1)
    stamp.scatter_reduce_(0, idx, e, reduce="amax")
    keep = stamp[idx] == e
    codes[idx[keep]] = val[keep]


def _plant(codes, g, device, consensus, fam, src_off, seg_len, dst, div, stamp, chunk=1 << 26):
    """Writes copies of family consensus sequences into the genome: copy i = consensus[fam[i]][src_off[i] : +seg_len[i]] at
    dst[i], every base substituted with probability div[i] (each copy mutates on its own)."""
    n = seg_len.numel()
    starts = torch.cumsum(seg_len, 0) - seg_len
    total = int(seg_len.sum())
    lo_seg = 0
    while lo_seg < n:  # whole copies per chunk
        hi_seg = int(torch.searchsorted(starts, starts[lo_seg] + chunk).item())
        hi_seg = max(hi_seg, lo_seg + 1)
        sl = slice(lo_seg, hi_seg)
        ln = seg_len[sl]
        m = int(ln.sum())
        seg = torch.repeat_interleave(torch.arange(hi_seg - lo_seg, device=device), ln)
        within = torch.arange(m, device=device) - (starts[sl] - starts[lo_seg])[seg]
        val = consensus[fam[sl][seg], src_off[sl][seg] + within]
        mut = torch.rand(m, generator=g, device=device) < div[sl][seg]
        val = torch.where(mut, (val + torch.randint(1, 4, (m,), generator=g, device=device, dtype=torch.uint8)) % 4, val)
        _put_last(codes, dst[sl][seg] + within, val, stamp)
        lo_seg = hi_seg
    return total


def make_genome(args, device, seed):
    """The synthetic stand-in for GRCh38 (which cannot be fetched offline), in HBM: contigs with a human-like length
    spread.  --genome human (default): a repeat landscape like the human one over a uniform random background —
    one short interspersed family at very high copy number (Alu-like), long interspersed families as truncated copies
    (L1-like), families with copy numbers from 3 to several hundred, segmental duplications, tandem repeats and
    low-complexity runs; copies diverge from their consensus by 1-15 %.  --genome uniform: round 1's genome (uniform
    random with 2000 x4 planted 1-kb repeats)."""
    g = torch.Generator(device=device).manual_seed(seed)
    total = int(args.genome_mbp * 1e6)
    w = torch.linspace(2.2, 0.5, args.contigs)
    lens = (w / w.sum() * total).long()
    lens = (lens // 4 * 4).clamp_(min=10000)
    G = int(lens.sum())
    codes = torch.randint(0, 4, (G,), generator=g, device=device, dtype=torch.uint8)
    stamp = torch.empty(G, dtype=torch.int32, device=device)  # (_put_last's scratch: 4 B per base, freed on return)
    if args.genome == "uniform":
        if args.repeats:
            rl = 1000
            unit_pos = torch.randint(0, codes.numel() - rl, (args.repeats,), generator=g, device=device)
            dst = torch.randint(0, codes.numel() - rl, (args.repeats, 3), generator=g, device=device)
            ar = torch.arange(rl, device=device)
            units = codes[(unit_pos[:, None] + ar[None, :])]
            for k in range(3):  # three extra copies of every family, ~1 % diverged
                u = units.clone()
                m = torch.rand(u.shape, generator=g, device=device) < 0.01
                u = torch.where(m, (u + torch.randint(1, 4, u.shape, generator=g, device=device, dtype=torch.uint8)) % 4, u)
                _put_last(codes, (dst[:, k][:, None] + ar[None, :]).reshape(-1), u.reshape(-1), stamp)
        return codes, [int(x) for x in lens], f"uniform random, {args.repeats} x4 1-kb repeat families at 1 % divergence"

    def rnd(n):
        return torch.rand(n, generator=g, device=device)

    def place(consensus, fam, src_off, seg_len, div):
        dst = (rnd(seg_len.numel()) * (G - int(seg_len.max()) - 1)).long()
        return _plant(codes, g, device, consensus, fam, src_off, seg_len, dst, div, stamp)

    planted = {}
    # (a) segmental duplications first (later, younger repeats land inside them too): 10-50 kb blocks copied 1-4 times at 1-3 %
    n = max(1, int(0.05 * G / 30_000 / 2.5))
    blk = (10_000 + rnd(n) * 40_000).long()
    src = (rnd(n) * (G - 60_000)).long()
    copies = 1 + (rnd(n) * 4).long()
    seg = torch.repeat_interleave(torch.arange(n, device=device), copies)
    seg_len = blk[seg]
    dst = (rnd(seg.numel()) * (G - 60_000)).long()
    starts = torch.cumsum(seg_len, 0) - seg_len
    tot = 0
    for lo in range(0, seg.numel(), 256):  # sources are genome ranges, not a consensus table: chunks of copies
        sl = slice(lo, min(lo + 256, seg.numel()))
        ln = seg_len[sl]
        m = int(ln.sum())
        s2 = torch.repeat_interleave(torch.arange(ln.numel(), device=device), ln)
        within = torch.arange(m, device=device) - (starts[sl] - starts[lo])[s2]
        val = codes[src[seg[sl]][s2] + within]
        mut = rnd(m) < (0.01 + 0.02 * rnd(ln.numel()))[s2]
        val = torch.where(mut, (val + torch.randint(1, 4, (m,), generator=g, device=device, dtype=torch.uint8)) % 4, val)
        _put_last(codes, dst[sl][s2] + within, val, stamp)
        tot += m
    planted["segmental duplications, 10-50 kb x1-4 at 1-3 %"] = tot
    # (b) families over a spectrum of copy numbers: 3 .. 300 copies (log-uniform), 1-3 kb, 2-10 % from the consensus
    nf = max(1, int(0.10 * G / (2000 * 64)))  # (a log-uniform 3..300 averages 64 copies)
    cons = torch.randint(0, 4, (nf, 3000), generator=g, device=device, dtype=torch.uint8)
    flen = (1000 + rnd(nf) * 2000).long()
    ncopy = torch.exp(rnd(nf) * (torch.log(torch.tensor(300.0)) - torch.log(torch.tensor(3.0))) + torch.log(torch.tensor(3.0))).long()
    fam = torch.repeat_interleave(torch.arange(nf, device=device), ncopy)
    planted["families of 3-300 copies, 1-3 kb at 2-10 %"] = place(cons, fam, torch.zeros_like(fam), flen[fam], 0.02 + 0.08 * rnd(fam.numel()))
    # (c) long interspersed families (L1-like): 40 consensus sequences of 6 kb, truncated copies (3' ends), 2-12 %
    nf = 40
    cons = torch.randint(0, 4, (nf, 6000), generator=g, device=device, dtype=torch.uint8)
    n = int(0.15 * G / 1500)
    fam = (rnd(n) * nf).long()
    seg_len = (300 + (rnd(n) ** 3) * 5700).long()  # most copies are short 3' fragments
    planted["40 long families (6 kb), truncated copies at 2-12 %"] = place(cons, fam, 6000 - seg_len, seg_len, 0.02 + 0.10 * rnd(n))
    # (d) one short family at very high copy number (Alu-like): 300 bp, 8-15 % from the consensus, poly-A tail
    cons = torch.randint(0, 4, (1, 300), generator=g, device=device, dtype=torch.uint8)
    cons[0, 280:] = 0
    n = int(0.10 * G / 300)
    fam = torch.zeros(n, dtype=torch.long, device=device)
    planted["one 300-bp family, %d copies at 8-15 %%" % n] = place(cons, fam, torch.zeros_like(fam), torch.full((n,), 300, device=device, dtype=torch.long),
                                                                   0.08 + 0.07 * rnd(n))
    # (e) tandem repeats and low complexity: units of 1-6 bp (microsatellites) and 10-60 bp (minisatellites), runs of 30-600 bp, 0-5 % impure
    n = int(0.03 * G / 150)
    unit = torch.where(rnd(n) < 0.7, 1 + (rnd(n) * 6).long(), 10 + (rnd(n) * 50).long())
    run = (30 + rnd(n) ** 2 * 570).long()
    ucons = torch.randint(0, 4, (n, 60), generator=g, device=device, dtype=torch.uint8)
    dst = (rnd(n) * (G - 700)).long()
    starts = torch.cumsum(run, 0) - run
    tot = 0
    for lo in range(0, n, 1 << 18):
        sl = slice(lo, min(lo + (1 << 18), n))
        ln = run[sl]
        m = int(ln.sum())
        s2 = torch.repeat_interleave(torch.arange(ln.numel(), device=device), ln)
        within = torch.arange(m, device=device) - (starts[sl] - starts[lo])[s2]
        val = ucons[sl][s2, within % unit[sl][s2]]
        mut = rnd(m) < (0.05 * rnd(ln.numel()))[s2]
        val = torch.where(mut, (val + torch.randint(1, 4, (m,), generator=g, device=device, dtype=torch.uint8)) % 4, val)
        _put_last(codes, dst[sl][s2] + within, val, stamp)
        tot += m
    planted["tandem repeats / low complexity"] = tot
    frac = sum(planted.values()) / G
    note = "human-like repeat landscape, %.0f %% of the bases planted as repeats: " % (100 * frac) + "; ".join(
        "%s (%.1f %%)" % (k, 100 * v / G) for k, v in planted.items())
    return codes, [int(x) for x in lens], note


def make_reads(codes, lens, n_pairs, rlen, seed, device, sub=0.005, ins=0.001, dele=0.001, paired=True):
    from mapcaller_amd import synth
    parts, o = [], 0
    for L in lens:
        parts.append(codes[o:o + L])
        o += L
    donor = synth.Genome([f"chr{i + 1}" for i in range(len(lens))], parts)
    bases, _ = synth.simulate_reads(donor, n_pairs, rlen, paired, seed, frag_mean=500, frag_sd=50, frag_min=300, frag_max=800,
                                    sub=sub, ins=ins, dele=dele, device=device, chunk=1 << 19, skip_head=3000)
    return bases  # uint8 ASCII [2 n_pairs (paired) or n_pairs, rlen]


def cpu_prepare(args, index, bases_sample):
    """The CPU baseline's inputs on disk — the index files as the reference loads them, the sample as FASTQ — in a directory the caller
    removes (cpu_run works on files only: it can run beside the GPU-only legs of this script)."""
    from mapcaller_amd import synth
    se = bool(args.single_end)
    step = 1 if se else 2
    tmp = tempfile.mkdtemp(prefix="mcx_cpu_")
    prefix = os.path.join(tmp, "idx")
    index.save(prefix)
    st = {"tmp": tmp, "prefix": prefix, "se": se, "step": step, "n_pairs": bases_sample.shape[0] // step, "alg": args.alg, "rlen": args.rlen, "level": args.cpu_level}
    # the -t 1 sample: ~20 s of the reference's clock (18-19 k reads/s against a human-sized index, 75 k against a bacterial one)
    t1 = args.cpu_t1_pairs if args.cpu_t1_pairs > 0 else (200_000 if args.genome_mbp > 100 else 1_500_000)
    st["t1_pairs"] = min(st["n_pairs"], t1)
    for tag, rows in (("r", bases_sample), ("t", bases_sample[:400]), ("s", bases_sample[:step * st["t1_pairs"]])):
        synth.write_fastq(os.path.join(tmp, tag + "1.fq"), rows, 0, step)
        if not se:
            synth.write_fastq(os.path.join(tmp, tag + "2.fq"), rows, 1, 2)
    return st


def usable_cpus():
    """The CPUs this process is given: the affinity mask cut by the cgroup's CPU-time share (cpu.max — the bench box says 1600000 100000 on a machine
    of 256 hardware threads: sixteen; what runs beyond the share is put to sleep for the rest of each 100 ms period).  mcx_cpus.h does the same for the library's host threads."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    share = None
    own = ""
    try:
        for line in open("/proc/self/cgroup"):
            if line.startswith("0::"):
                own = line[3:].strip()
    except OSError:
        pass
    p = own
    while True:
        for f in ("/sys/fs/cgroup" + p + "/cpu.max",):
            try:
                q, per = open(f).read().split()[:2]
                if q != "max" and float(per) > 0:
                    share = min(share, float(q) / float(per)) if share else float(q) / float(per)
            except (OSError, ValueError):
                pass
        if p in ("", "/"):
            break
        p = p[:p.rfind("/")]
    try:
        q, per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            share = min(share, q / per) if share else q / per
    except (OSError, ValueError):
        pass
    if share:
        n = min(n, max(1, int(share + 0.5)))
    return max(1, n)


def cpu_run(st):
    """The CPU path on this box's host cores, on a bounded sample of the same workload, index load excluded (the reference
    starts its clock after loading, main.cpp:376).  At -t <the CPUs the box gives, usable_cpus()>: with `-sam <file>` (the reference then formats every line and
    pushes it through fprintf under its OutputLock, ReadMapping.cpp:536-560) and — level "two" / "full" — without any output (`-no_vcf`, no
    `-sam`: bSAMoutput stays false, :536 is skipped): `mapping_only`, the like-for-like figure beside `value`, which times kernels and writes
    no text either.  Level "full" (the default) adds a -t 1 run on about 20 s worth of pairs (--cpu-t1-pairs); for the GRCh38-sized
    index every run of the reference spends ~45 s loading it: three runs, about three minutes of the default line."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "MapCaller")
    port_bin = os.path.join(ROOT, "oracle", "mcx_oracle")
    cores = usable_cpus()  # (-t <what the box gives>: at -t <hardware threads> under a CPU-time share the reference spends most of each period asleep)
    tmp, prefix, se, step, n_pairs, alg = st["tmp"], st["prefix"], st["se"], st["step"], st["n_pairs"], st["alg"]
    fq = lambda tag: (os.path.join(tmp, tag + "1.fq"), os.path.join(tmp, tag + "2.fq"))
    if os.path.exists(ref_bin):
        kind = "reference"
        def run(a, b, threads=cores, sam=True):
            cmd = [ref_bin, "-i", prefix, "-f", a] + ([] if se else ["-f2", b]) + ["-alg", alg] + (["-sam", os.path.join(tmp, "o.sam")] if sam else []) + \
                  ["-no_vcf", "-t", str(threads), "-log", os.path.join(tmp, "job.log")]
            t0 = time.perf_counter()
            r = subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
            wall = time.perf_counter() - t0
            # the reference's own clock starts after the index is loaded (main.cpp:376) and prints whole seconds
            m = re.findall(r"have been processed in (\d+) seconds", r.stderr)
            return wall, (int(m[-1]) if m else None)
    elif os.path.exists(port_bin):
        kind = "port"
        def run(a, b, threads=cores, sam=True):
            cmd = [port_bin, "-i", prefix, "-f", a] + ([] if se else ["-f2", b]) + ["-alg", alg, "-sam", os.path.join(tmp, "o.sam") if sam else "/dev/null", "-t", str(threads)]
            t0 = time.perf_counter()
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            return time.perf_counter() - t0, None
    else:
        return None
    t_load = [None]

    def rate(sam, files, reps=1):
        t_full, own = run(files[0], files[1], sam=sam)
        if own is not None and own >= 5:  # (whole seconds: a run of a few seconds cannot be read off it)
            dt, how = float(own), f"the reference's own clock (starts after the index load; whole seconds): {own} s of {t_full:.1f} s wall"
        else:
            if t_load[0] is None:
                t_load[0] = run(*fq("t"), sam=False)[0]  # 200 pairs: index load + start-up (what a wall clock has to be corrected by)
            dt, how = max(t_full - t_load[0], 1e-3), f"wall {t_full:.2f} s minus {t_load[0]:.2f} s of index load and start-up" + (f" (its own clock: {own} s)" if own is not None else "")
        return round(reps * step * n_pairs / dt, 1), how
    v_sam, how_sam = rate(True, fq("r"))
    what = f"{n_pairs} {'reads' if se else 'pairs'} x {st['rlen']} bp of the same synthetic workload, -t {cores} -alg {alg}"
    out = {"value": v_sam, "unit": "reads/s", "cores": cores, "hardware_threads": os.cpu_count(), "kind": kind, "sample": f"{what} -sam (file) -no_vcf; {how_sam}",
           "note": "-t = the CPUs the box gives this process (usable_cpus(): affinity and the cgroup's CPU-time share; the machine has %d hardware threads) — rounds 1-4 ran it at "
                   "-t <hardware threads>, which a CPU-time share turns into sleeping; its clock prints whole seconds (+-1 s of the figure above)" % (os.cpu_count() or 1)}
    if kind == "reference" and st["level"] in ("two", "full"):
        v_map, how_map = rate(False, fq("r"))
        out["mapping_only"] = {"value": v_map, "unit": "reads/s", "cores": cores,
                               "sample": f"{what}, -no_vcf and no -sam: mapping alone, no SAM text (ReadMapping.cpp:536 skipped); {how_map}"}
    if kind == "reference" and st["level"] == "full":
        n1 = st["t1_pairs"]
        _, own1 = run(*fq("s"), threads=1)
        if own1:
            out["single_thread"] = {"value": round(step * n1 / own1, 1), "unit": "reads/s", "cores": 1,
                                    "sample": f"{n1} {'reads' if se else 'pairs'}, -t 1, -sam (file), the reference's own clock (whole seconds): {own1} s"}
    return out


def cpu_baseline(args, index, bases_sample):
    st = cpu_prepare(args, index, bases_sample)
    try:
        return cpu_run(st)
    finally:
        shutil.rmtree(st["tmp"], ignore_errors=True)


def launch_ranks(args):
    """--gpus N from a plain `python bench.py`: N ranks as fresh child processes (torch.distributed.run), started
    before this process makes any GPU call; the children carry WORLD_SIZE and run main() below."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    have = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    if have < args.gpus and not os.environ.get("MCX_BENCH_SHARE_GPU"):
        sys.exit(f"bench.py --gpus {args.gpus}: this node shows {have} GPU(s)")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


class Trajectory:
    """The run's insert-size state across the ranks (N > 1): every step is one round of the input stream — rank r maps
    batch N*step + r — and the ranks walk ONE avgDist trajectory over the round (ReadMapping.cpp:462, :538-539), as
    mcx_map_files_ex does for real inputs.  What crosses the ranks is three numbers a rank and exchange — pairs re-run, proper
    pairs, their summed distance (mcx_batch_totals) —: the walk has a closed form, so a rank checks its own chunks on the device from
    the round's state plus the totals of the ranks before it (mcx_batch_check) and re-runs the pairs whose estimate moved, until no
    rank re-ran any.  (Round 4 sent every chunk's sums — 640 KB a rank at 4 M pairs — and walked them on the host.)"""

    def __init__(self, dist, dev, world, rank, n_chunks):
        self.dist, self.dev, self.world, self.rank, self.nc = dist, dev, world, rank, n_chunks
        self.state = [1000, 0, 0]
        self.reads = 0
        self.cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")  # (gloo: several ranks sharing one GPU in tests)
        self.h_msg = torch.zeros(3, dtype=torch.int64).pin_memory()
        self.h_all = torch.zeros(world * 3, dtype=torch.int64).pin_memory()  # (flat: what all_gather_into_tensor takes on every backend)
        self.msg = torch.zeros(3, dtype=torch.int64, device=self.cdev)
        self.all = torch.zeros(world * 3, dtype=torch.int64, device=self.cdev)
        self.exchanges = 0
        self.host_s = 0.0  # time between the end of a batch's kernels and the start of the next ones: staging, all-gather, bookkeeping

    def step(self, mapper, d_bases, d_off, n_reads, d_aln, d_cig):
        from mapcaller_amd import api
        est0 = int(float(self.state[0]) * 1.5)
        mapper.batch_begin(d_bases, d_off, n_reads, True, est0, self.reads + self.rank * n_reads, d_aln, d_cig)
        n_redo = -1
        hm = self.h_msg.numpy()
        for it in range(64):
            pairs, dsum = mapper.batch_totals()
            t0 = time.perf_counter()
            hm[0], hm[1], hm[2] = n_redo, pairs, dsum
            if self.cdev.type == "cuda":
                self.msg.copy_(self.h_msg, non_blocking=True)
                self.dist.all_gather_into_tensor(self.all, self.msg)
                self.h_all.copy_(self.all, non_blocking=True)
                torch.cuda.current_stream().synchronize()
            else:
                self.dist.all_gather_into_tensor(self.h_all, self.h_msg)
            self.exchanges += 1
            h = self.h_all.numpy().reshape(self.world, 3)
            before = [self.state[0], self.state[1] + int(h[:self.rank, 1].sum()), self.state[2] + int(h[:self.rank, 2].sum())]
            self.host_s += time.perf_counter() - t0
            if it > 0 and not h[:, 0].any():
                break
            n_redo = mapper.batch_check(before, self.rank == 0)
        else:
            raise RuntimeError("avgDist replay did not converge")
        api.avg_advance(self.state, int(h[:, 1].sum()), int(h[:, 2].sum()), self.world * self.nc)
        self.reads += self.world * n_reads
        mapper.batch_end()


def main():
    args = parse()
    try:  # (torch's CPU pool sized for the CPUs the process is given, not the machine's 256 hardware threads: what it does on the host here is copies and small index arithmetic)
        torch.set_num_threads(max(1, min(usable_cpus(), 16)))
    except Exception:
        pass
    launch_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    local = local % n_dev if os.environ.get("MCX_BENCH_SHARE_GPU") else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("MCX_FORCE_DIST"):
        import torch.distributed as dist_
        dist = dist_
        backend = os.environ.get("MCX_BENCH_BACKEND", "nccl")  # "gloo" lets test ranks share one GPU (RCCL refuses that)
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    from mapcaller_amd import api
    if not os.path.exists(api.LIB_PATH):  # a checkout without the built artefacts: build them (never a CPU fallback)
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if dist:
            dist.barrier()

    # ---- set-up (not timed): genome, index, reads ------------------------------------------------
    codes, lens, genome_note = make_genome(args, dev, seed=1234)
    native_dir = None
    if args.native_boundary and args.pcie_steps > 0 and world == 1 and not args.single_end and os.path.isdir("/dev/shm"):
        try:  # (a tmpfs too small for the genome: the leg from the other process is left out, nothing else)
            native_dir = tempfile.mkdtemp(prefix="mcx_boundary_", dir="/dev/shm")  # what the other process's run is made of: the genome now, the packed batches by pcie_inclusive
            import atexit
            atexit.register(shutil.rmtree, native_dir, True)
            codes.cpu().numpy().tofile(os.path.join(native_dir, "genome.u8"))
            json.dump({"chr_lens": [int(x) for x in lens], "alg": args.alg, "rlen": args.rlen, "reads": 2 * args.batch_pairs, "steps": args.pcie_steps, "full_sa": int(args.full_sa)},
                      open(os.path.join(native_dir, "meta.json"), "w"))
            args.native_dir = native_dir
        except OSError:
            native_dir = None
    t0 = time.perf_counter()
    index = api.Index.from_codes(codes.data_ptr(), lens, device=local, full_sa=int(args.full_sa))
    t_index = time.perf_counter() - t0
    index_gb = round(index.hbm_bytes / 1e9, 2)  # (before the -vcf leg gives the pair records back)
    n_steps = args.warmup + args.steps
    paired = not args.single_end
    reads_per_step = (2 if paired else 1) * args.batch_pairs
    mapper = api.Mapper(index, alg=args.alg, max_read_len=max(256, args.rlen), max_batch_reads=reads_per_step)
    batches = []
    for s in range(n_steps):
        b = make_reads(codes, lens, args.batch_pairs, args.rlen, seed=1000 * (rank + 1) + s, device=dev, sub=args.sub, ins=args.ins, dele=args.dele, paired=paired)
        batches.append(b.reshape(-1).contiguous())
    off = (torch.arange(reads_per_step + 1, device=dev, dtype=torch.int64) * args.rlen).to(torch.uint32)
    cpu_pairs = args.cpu_pairs
    if cpu_pairs < 0:  # ~20 s of the reference's clock: 15 k reads/s a core up to ~16 cores, ~240 k reads/s beyond (its locks), bounded by one batch
        cpu_pairs = int(min(args.batch_pairs, max(50_000, min(usable_cpus() * 15_000, 240_000) * 20 // 2)))
    sample = None
    if rank == 0 and world == 1 and cpu_pairs:
        per = 2 if paired else 1
        parts = [batches[0].reshape(reads_per_step, args.rlen)[: per * cpu_pairs].cpu()]
        have = parts[0].shape[0] // per
        k = 0
        while have < cpu_pairs:  # (a sample larger than a batch — config 2's million reads are a third of a second of CPU work: further batches of the same kind)
            m = min(args.batch_pairs, cpu_pairs - have)
            parts.append(make_reads(codes, lens, m, args.rlen, seed=900_000 + k, device=dev, sub=args.sub, ins=args.ins, dele=args.dele, paired=paired).reshape(per * m, args.rlen).cpu())
            have += m
            k += 1
        sample = torch.cat(parts) if len(parts) > 1 else parts[0]
        del parts
    del codes
    d_aln = torch.empty(reads_per_step * 64, dtype=torch.uint8, device=dev)
    d_cig = torch.empty(api.cigar_pool_words(reads_per_step), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    traj = Trajectory(dist, dev, world, rank, (args.batch_pairs + 99) // 100) if (world > 1 and paired) else None

    def step(i):
        if traj:
            traj.step(mapper, batches[i].data_ptr(), off.data_ptr(), reads_per_step, d_aln.data_ptr(), d_cig.data_ptr())
        else:
            mapper.map_batch_dev(batches[i].data_ptr(), off.data_ptr(), reads_per_step, paired, d_aln.data_ptr(), d_cig.data_ptr())

    # (the set-up above leaves host threads behind that still spin — torch's CPU pool after the .cpu() copies — and the box gives the process a CPU-time SHARE:
    #  a group that has spent it sleeps until the next 100 ms period, this thread with it; let the period turn before the clock starts)
    time.sleep(0.3)
    for i in range(args.warmup):
        step(i)
    before = mapper.stats.as_dict()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_steps):
        step(i)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist:
        from mapcaller_amd import dist as mdist_
        dt = mdist_.max_over_ranks(dt, dev)
    after = mapper.stats.as_dict()
    d = {k: after[k] - before[k] for k in after}

    # ---- the same batches from pinned host memory, records back to pinned host memory (not `value`) ----------
    pcie = None
    if args.pcie_steps > 0 and paired:
        try:
            pcie = pcie_inclusive(args, mapper, batches, reads_per_step, dist, dev)
        except Exception as e:
            pcie = {"error": str(e)[:300]}

    # (the legs below bring contexts and buffers of their own: the timed region's context, its device slots and all but one batch make room)
    keep = batches[min(args.warmup, len(batches) - 1)]
    kept = batches[args.warmup:args.warmup + max(1, args.file_batches)] or [keep]  # the file leg's reads
    vcf_batches = (batches[args.warmup:args.warmup + 5] or [keep]) if (args.vcf_reduce == 1 or (args.vcf_reduce < 0 and world > 1)) else []  # the -vcf leg's
    del batches[:]
    batches.append(keep)
    mapper.close()
    mapper = None
    torch.cuda.empty_cache()

    # ---- files in, SAM out (the CLI's path), batches of the timed region as FASTQ in tmpfs ------------------------------------
    f2f = None
    if args.file_steps > 0 and paired and world == 1:
        try:
            f2f = file_to_file(args, index, kept, reads_per_step)
        except Exception as e:
            f2f = {"error": str(e)[:300]}
    del kept[:]
    torch.cuda.empty_cache()

    # ---- the bulk exchange of a -vcf run (not timed): profile of one batch, RCCL reduce over the ranks ------
    vcf = None
    do_vcf = args.vcf_reduce == 1 or (args.vcf_reduce < 0 and world > 1)
    if do_vcf:
        try:
            vcf = vcf_leg(args, index, mapper, vcf_batches, off, d_aln, d_cig, reads_per_step, 1, d, dist, dev, rank, world)
            mapper = None
        except Exception as e:  # never lose the bench line to the optional section
            vcf = {"error": str(e)[:300]}
        del vcf_batches[:]
        torch.cuda.empty_cache()

    if rank == 0:
        total_reads = reads_per_step * args.steps * world
        out = {
            "metric": "reads/sec (150 bp PE vs GRCh38) at 1/2/4/8 MI355X; SAM CIGAR bit-exact",
            "value": round(total_reads / dt, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64/int32", "dtype_note": "FM-index intervals and rank counts in u64 / u32; the DP recurrences in 32-bit registers holding ksw2's int8-range differences "
                                                "and nw's doubled scores, traceback flags packed 2 / 4 bits per cell",
            "data": "synthetic",
            "config": {"workload": f"synthetic GRCh38-sized genome, {args.genome_mbp:.0f} Mbp ({args.contigs} contigs, {genome_note}; GRCh38 itself is unavailable offline), "
                                   f"{args.batch_pairs} {'pairs' if paired else 'reads'} x {args.rlen} bp {'PE' if paired else 'SE'} per step per GPU (sub {args.sub}, ins {args.ins}, del {args.dele} per base), -alg {args.alg}",
                       "reads_per_step_per_gpu": reads_per_step, "full_sa_in_hbm": bool(args.full_sa), "pair_records_in_hbm": args.full_sa >= 2, "index_build_s": round(t_index, 2),
                       "index_hbm_gb": index_gb,
                       "multi_gpu": None if world == 1 else f"one process per GPU, index replicated, rank r maps batch {world}*step + r; one avgDist trajectory per step over the ranks' batches: "
                                                            f"{traj.exchanges} exchanges in {n_steps} steps, inside the timed region (all-gather of three numbers a rank over RCCL: pairs re-run, proper pairs, summed distance)",
                       "multi_gpu_host_ms_per_step": None if traj is None else round(1000 * traj.host_s / n_steps, 3)},
            "roofline": roofline(args, d, total_reads / dt / world),
            "per_read": {"fm_ext_steps": round(d["fm_ext_steps"] / max(d["reads"], 1), 2), "fm_blocks": round(d["fm_blocks"] / max(d["reads"], 1), 2),
                         "sa_hits": round(d["sa_hits"] / max(d["reads"], 1), 3), "dp_jobs": round(d["dp_jobs"] / max(d["reads"], 1), 4),
                         "mapped_frac": round(d["mapped"] / max(d["reads"], 1), 4)},
            "stage_ms_per_step": {k[3:]: round(d[k] / args.steps, 3) for k in d if k.startswith("ms_")},
            "tier1_pairs": d["tier1_pairs"], "replayed_pairs": d["replayed_pairs"], "halved_selections": d["halved_selections"], "simple_pairs": d.get("simple_pairs", 0),
            "dp": {"jobs": d["dp_jobs"], "cells": d["dp_cells"], "ms_per_step": round(d["ms_dp"] / args.steps, 3),
                   "gcups": round(d["dp_cells"] / max(d["ms_dp"], 1e-9) / 1e6, 2),
                   "note": "cell updates of all gapped-extension problems (query x target) over the time of the DP stage (the lists' kernels on three streams)"},
        }
        if pcie is not None:
            out["value_pcie_inclusive"] = pcie
        if f2f is not None:
            out["value_file_to_file"] = f2f
        if vcf is not None:
            out["vcf_reduce"] = vcf
        # (the reference at -t <all cores> beside a leg of this script starves that leg's host threads — tried: the -vcf leg's host passes took
        #  forty times as long — so the baseline has the box to itself: two runs, most of each the reference loading the index)
        if sample is not None:
            try:
                out["cpu_baseline"] = cpu_baseline(args, index, sample)
            except Exception as e:  # the baseline must never take the bench line down
                out["cpu_baseline"] = {"error": str(e)[:200]}
        nd = getattr(args, "native_dir", None)
        if (args.second_genome or args.other_configs or nd) and world == 1:
            try:
                if mapper is not None:
                    mapper.close()
                del batches[:]
                index.close()
                del d_aln, d_cig
                torch.cuda.empty_cache()
            except Exception:
                pass
            if nd and pcie is not None and "error" not in pcie:
                try:
                    pcie["system_runtime"] = native_boundary(args, pcie)
                except Exception as e:
                    pcie["system_runtime"] = {"error": str(e)[:200]}
            if args.second_genome:
                try:
                    out["other_genome"] = other_genome(args)
                except Exception as e:
                    out["other_genome"] = {"error": str(e)[:200]}
            if args.other_configs:
                out["other_configs"] = other_configs(args)
        emit(out, args)
    if dist:
        # (everything is measured and printed: leaving must not fail the run.  Over gloo — test ranks sharing one GPU — a rank that is through the barrier and closes its
        #  sockets while its neighbours are still inside it gives them "Connection closed by peer": seen once in three 8-rank runs of the suite.  A rank waits a moment
        #  before it leaves, and a peer that has already left is no error here.)
        try:
            dist.barrier()
            time.sleep(0.5 if dist.get_backend() == "gloo" else 0.0)
            dist.destroy_process_group()
        except RuntimeError as e:
            print(f"bench.py: rank {rank} left behind a peer that had already gone ({str(e)[:120]})", file=sys.stderr)


if __name__ == "__main__":
    main()
