#!/usr/bin/env python3
"""A markdown table of a rocprofv3 summary's heaviest kernels (scripts/summarize_profile.py's output): time a batch, the longest launch, its HBM rate against 8 TB/s,
L2 misses a second, and how its wavefronts spent their cycles — the table DESIGN.md §3 carries for BASELINE config 5.
    python scripts/kernel_table.py profiles/round6/summary_cfg5.json [batches]"""
import json, sys
o = json.load(open(sys.argv[1]))
batches = float(o.get("batches_mapped_by_the_pmc_runs") or 3)  # the PMC passes: a warm-up batch and two steps
trace_batches = float(sys.argv[2]) if len(sys.argv) > 2 else batches + 1  # the kernel trace: one step more (scripts/collect_profile.sh)
skip = ("k_pair_keys", "k_scatter_rank", "k_pack_bwt", "k_pair_codes", "k_mark_heads", "k_init_keys", "k_pair_records", "k_build_ktab", "k_bucket", "k_sample_sa", "k_build_rank", "k_find_primary", "Cijk", "at::", "void at::", "__amd", "k_fill", "k_make", "k_mut", "k_synth")
rows = []
for k, v in o["kernel_trace"].items():
    if any(k.startswith(s) or (" " + s) in k for s in skip) or k not in o["pmc"]:
        continue
    p = o["pmc"][k]
    h = o.get("hbm_traffic", {}).get(k, {})
    full = lambda c: float((p.get(c) or {}).get("full_batch_mean") or 0.0)  # per full-batch launch (the PMC passes' launches, serialised by the profiler)
    tot = lambda c: float((p.get(c) or {}).get("total") or 0.0)
    t = float(v.get("full_batch_avg_us") or 0.0) / 1e6  # the same launches in the kernel trace (beside whatever shared the chip with them)
    ms_batch = v["avg_us"] * v["launches"] / trace_batches / 1000.0
    if ms_batch < 1.0 or t <= 0:
        continue
    gb = (float(h.get("hbm_read_bytes_per_launch") or 0) + float(h.get("hbm_write_bytes_per_launch") or 0)) / 1e9
    wave, wait, act = tot("SQ_WAVE_CYCLES"), tot("SQ_WAIT_ANY"), tot("SQ_ACTIVE_INST_ANY")
    rows.append((ms_batch, k, float(v["full_batch_avg_us"]) / 1000.0, v["max_us"] / 1000.0, gb, gb / t / 1000.0, full("TCC_MISS_sum") / t / 1e9,
                 wait / wave if wave else 0.0, act / wave if wave else 0.0, tot("SQ_INSTS_VALU") / batches / 1e9))
print("| kernel | ms a batch (all launches) | a full-batch launch: ms (longest) | its HBM bytes, GB | TB/s (of 8) | L2 misses, G/s | waiting | issuing | vector instructions a batch, G |")
print("|---|---|---|---|---|---|---|---|---|")
for r in sorted(rows, reverse=True):
    print("| `%s` | %.1f | %.1f (%.1f) | %.2f | %.2f (%.2f) | %.1f | %.2f | %.2f | %.1f |" % (r[1][:40], r[0], r[2], r[3], r[4], r[5], r[5] / 8.0, r[6], r[7], r[8], r[9]))
