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
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
PMC_SUMMARY = os.path.join(ROOT, "profiles", "round1", "summary_3gbp_r1d.json")


def pmc_profile(args):
    """What the committed rocprofv3 --pmc passes of this same command measured for one k_seed launch
    (counters cannot be read from inside the run): HBM bytes (FETCH_SIZE + WRITE_SIZE) and L2
    requests (TCC_HIT + TCC_MISS).  Only returned when the workload is the one those passes profiled."""
    if (args.genome_mbp, args.batch_pairs, args.rlen, args.alg) != (3100.0, 4_000_000, 150, "ksw2"):
        return None
    try:
        with open(PMC_SUMMARY) as fh:
            s = json.load(fh)
        t, p = s["hbm_traffic"]["k_seed"], s["pmc"]["k_seed"]
        per_kernel = {k: int(v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"]) for k, v in s["hbm_traffic"].items()
                      if k in ("k_pack_reads", "k_seed", "k_cluster", "k_build", "k_finish")}
        return {"per_kernel": per_kernel, "traffic": int(t["hbm_read_bytes_per_launch"] + t["hbm_write_bytes_per_launch"]),
                "l2_requests": int(p["TCC_HIT_sum"]["full_batch_mean"] + p["TCC_MISS_sum"]["full_batch_mean"]),
                "wait_frac": round(p["SQ_WAIT_ANY"]["full_batch_mean"] / p["SQ_WAVE_CYCLES"]["full_batch_mean"], 3)}
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None


def path_roofline(d, args, reads_per_s):
    reads = max(d["reads"], 1)
    e, h = d["fm_ext_steps"] / reads, d["sa_hits"] / reads
    b_read = 64.0 * (1.107 * e + 31.0 * h) + 8.0 * h + args.rlen + args.rlen / 4.0
    gbs = reads_per_s * b_read / 1e9
    return {"bytes_per_read": round(b_read, 1), "achieved": round(gbs, 1), "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "bound_reads_per_s": round(HBM_PEAK_GBS * 1e9 / b_read, 1)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=3100.0, help="synthetic genome size (Mbp)")
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--batch-pairs", type=int, default=4_000_000, help="read pairs per step and per GPU")
    ap.add_argument("--rlen", type=int, default=150)
    ap.add_argument("--sub", type=float, default=0.005, help="per-base substitution rate of the simulated reads")
    ap.add_argument("--ins", type=float, default=0.001, help="per-base insertion rate")
    ap.add_argument("--dele", type=float, default=0.001, help="per-base deletion rate")
    ap.add_argument("--alg", default="ksw2", choices=["nw", "ksw2"])
    ap.add_argument("--full-sa", type=int, default=1, help="keep every suffix-array entry in HBM")
    ap.add_argument("--cpu-pairs", type=int, default=-1,
                    help="pairs of the CPU-baseline sample (0 = skip, -1 = about 20 s of work for this host's core count)")
    ap.add_argument("--repeats", type=int, default=2000, help="planted dispersed repeat families")
    ap.add_argument("--vcf-reduce", type=int, default=1,
                    help="after the timed region: accumulate the -vcf alignment profile of one batch and sum it over the "
                         "ranks with RCCL (1 = yes, 0 = no, -1 = only when more than one GPU)")
    return ap.parse_args()


def make_genome(args, device, seed):
    """Uniform random contigs (human-like length spread) with planted 1 kb repeats, in HBM."""
    g = torch.Generator(device=device).manual_seed(seed)
    total = int(args.genome_mbp * 1e6)
    w = torch.linspace(2.2, 0.5, args.contigs)
    lens = (w / w.sum() * total).long()
    lens = (lens // 4 * 4).clamp_(min=10000)
    codes = torch.randint(0, 4, (int(lens.sum()),), generator=g, device=device, dtype=torch.uint8)
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
            codes[(dst[:, k][:, None] + ar[None, :]).reshape(-1)] = u.reshape(-1)
    return codes, [int(x) for x in lens]


def make_reads(codes, lens, n_pairs, rlen, seed, device, sub=0.005, ins=0.001, dele=0.001):
    from mapcaller_amd import synth
    parts, o = [], 0
    for L in lens:
        parts.append(codes[o:o + L])
        o += L
    donor = synth.Genome([f"chr{i + 1}" for i in range(len(lens))], parts)
    bases, _ = synth.simulate_reads(donor, n_pairs, rlen, True, seed, frag_mean=500, frag_sd=50, frag_min=300, frag_max=800,
                                    sub=sub, ins=ins, dele=dele, device=device, chunk=1 << 19, skip_head=3000)
    return bases  # uint8 ASCII [2 n_pairs, rlen]


def cpu_baseline(args, index, bases_sample):
    """The CPU path on this box's host cores, on a bounded sample of the same workload, index
    load excluded (the reference starts its clock after loading, main.cpp:376)."""
    from mapcaller_amd import synth
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "MapCaller")
    port_bin = os.path.join(ROOT, "oracle", "mcx_oracle")
    cores = os.cpu_count() or 1
    n_pairs = bases_sample.shape[0] // 2
    with tempfile.TemporaryDirectory() as tmp:
        prefix = os.path.join(tmp, "idx")
        index.save(prefix)
        f1, f2 = os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
        t1, t2 = os.path.join(tmp, "t1.fq"), os.path.join(tmp, "t2.fq")
        synth.write_fastq(f1, bases_sample, 0, 2)
        synth.write_fastq(f2, bases_sample, 1, 2)
        synth.write_fastq(t1, bases_sample[:400], 0, 2)
        synth.write_fastq(t2, bases_sample[:400], 1, 2)
        if os.path.exists(ref_bin):
            kind = "reference"
            def run(a, b, threads=cores):
                cmd = [ref_bin, "-i", prefix, "-f", a, "-f2", b, "-alg", args.alg, "-sam", os.path.join(tmp, "o.sam"), "-no_vcf", "-t", str(threads), "-log", os.path.join(tmp, "job.log")]
                t0 = time.perf_counter()
                r = subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
                wall = time.perf_counter() - t0
                # the reference's own clock starts after the index is loaded (main.cpp:376) and prints whole seconds
                m = re.findall(r"have been processed in (\d+) seconds", r.stderr)
                return wall, (int(m[-1]) if m else None)
        elif os.path.exists(port_bin):
            kind = "port"
            def run(a, b):
                cmd = [port_bin, "-i", prefix, "-f", a, "-f2", b, "-alg", args.alg, "-sam", os.path.join(tmp, "o.sam"), "-t", str(cores)]
                t0 = time.perf_counter()
                subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                return time.perf_counter() - t0, None
        else:
            return None
        t_full, own = run(f1, f2)
        if own is not None and own >= 3:
            dt, how = float(own), f"the reference's own clock (starts after the index load): {own} s of {t_full:.1f} s wall"
        else:
            t_load, _ = run(t1, t2)    # 200 pairs: index load + start-up
            dt, how = max(t_full - t_load, 1e-3), f"wall {t_full:.1f}s minus {t_load:.1f}s index load"
        out = {"value": round(2 * n_pairs / dt, 1), "unit": "reads/s", "cores": cores, "kind": kind,
               "sample": f"{n_pairs} pairs x {args.rlen} bp of the same synthetic workload, -t {cores} -alg {args.alg} -sam (file) -no_vcf; {how}"}
        if kind == "reference":  # SURVEY 8d also asks for -t 1: a smaller sample, the reference's own clock again
            n1 = min(n_pairs, 75_000)
            s1, s2 = os.path.join(tmp, "s1.fq"), os.path.join(tmp, "s2.fq")
            synth.write_fastq(s1, bases_sample[:2 * n1], 0, 2)
            synth.write_fastq(s2, bases_sample[:2 * n1], 1, 2)
            _, own1 = run(s1, s2, threads=1)
            if own1:
                out["single_thread"] = {"value": round(2 * n1 / own1, 1), "unit": "reads/s", "cores": 1,
                                        "sample": f"{n1} pairs, -t 1, the reference's own clock: {own1} s"}
        return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("MCX_FORCE_DIST"):
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=dev)
    from mapcaller_amd import api
    if not os.path.exists(api.LIB_PATH):  # a checkout without the built artefacts: build them (never a CPU fallback)
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if dist:
            dist.barrier()

    # ---- set-up (not timed): genome, index, reads ------------------------------------------------
    codes, lens = make_genome(args, dev, seed=1234)
    t0 = time.perf_counter()
    index = api.Index.from_codes(codes.data_ptr(), lens, device=local, full_sa=bool(args.full_sa))
    t_index = time.perf_counter() - t0
    n_steps = args.warmup + args.steps
    reads_per_step = 2 * args.batch_pairs
    mapper = api.Mapper(index, alg=args.alg, max_read_len=max(256, args.rlen), max_batch_reads=reads_per_step)
    batches = []
    for s in range(n_steps):
        b = make_reads(codes, lens, args.batch_pairs, args.rlen, seed=1000 * (rank + 1) + s, device=dev, sub=args.sub, ins=args.ins, dele=args.dele)
        batches.append(b.reshape(-1).contiguous())
    off = (torch.arange(reads_per_step + 1, device=dev, dtype=torch.int64) * args.rlen).to(torch.uint32)
    cpu_pairs = args.cpu_pairs
    if cpu_pairs < 0:  # ~20 s at ~15 k reads/s/core, bounded by one batch
        cpu_pairs = int(min(args.batch_pairs, max(50_000, (os.cpu_count() or 1) * 15_000 * 20 // 2)))
    sample = batches[0].reshape(reads_per_step, args.rlen)[: 2 * cpu_pairs].cpu() if (rank == 0 and world == 1 and cpu_pairs) else None
    del codes
    d_aln = torch.empty(reads_per_step * 64, dtype=torch.uint8, device=dev)
    d_cig = torch.empty(reads_per_step * api.CIGAR_STRIDE, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step(i):
        mapper.map_batch_dev(batches[i].data_ptr(), off.data_ptr(), reads_per_step, True, d_aln.data_ptr(), d_cig.data_ptr())

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

    # ---- the one exchange of a -vcf run (not timed): profile of one batch, RCCL sum over the ranks -----
    vcf = None
    do_vcf = args.vcf_reduce == 1 or (args.vcf_reduce < 0 and world > 1)
    if do_vcf:
        try:
            from mapcaller_amd import dist as mdist
            G = index.genome_size
            # this leg keeps ten planes of the genome (124 GB at 3.1 Gbp) and per-read alignment detail in HBM:
            # the timed region's context and all but one batch make room, the batch is mapped in slices
            last = batches[n_steps - 1]
            del batches[:]
            mapper.close()
            slice_reads = min(reads_per_step, 2_000_000)
            mapper = api.Mapper(index, alg=args.alg, max_read_len=max(256, args.rlen), max_batch_reads=slice_reads)
            planes = torch.zeros((10, G), dtype=torch.int32, device=dev)
            mapper.profile_attach(planes.data_ptr())
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for lo in range(0, reads_per_step, slice_reads):
                n = min(slice_reads, reads_per_step - lo)
                mapper.map_batch_dev(last.data_ptr() + lo * args.rlen, off.data_ptr(), n, True, d_aln.data_ptr(), d_cig.data_ptr())
            torch.cuda.synchronize()
            t_acc = time.perf_counter() - t1
            sparse = mapper.profile_sparse_raw()
            t2 = time.perf_counter()
            planes, merged = mdist.reduce_profile(planes, sparse)
            torch.cuda.synchronize()
            t_red = time.perf_counter() - t2
            mapper.profile_finalize(planes.data_ptr())
            tot = mdist.sum_over_ranks([d["pairs"], d["pair_dist_sum"], d["pair_len_sum"]], dev)
            vcf = {"profile_batch_ms": round(1000 * t_acc, 2), "allreduce_ms": round(1000 * t_red, 2),
                   "allreduce_gb": round(planes.numel() * 4 / 1e9, 2), "sparse_records": len(merged),
                   "covered_positions": int(((planes[0] | planes[1] | planes[2] | planes[3]) > 0).sum().item())}
            if rank == 0:  # VariantCalling() runs once, on the reduced profile
                with tempfile.TemporaryDirectory() as tmp:
                    vs = index.call_variants(planes.data_ptr(), merged, tot[0], tot[1], tot[2], os.path.join(tmp, "bench.vcf"),
                                             ref_name="synthetic", cmdline="bench.py")
                # both scans stream the planes once: 16 B (k_vc_depth) and 20 B + 2-bit base + depth word (k_vc_scan) per position
                vcf["call_variants"] = {"ms_total": round(vs["ms_total"], 2), "k_vc_depth_ms": round(vs["ms_depth"], 3),
                                        "k_vc_scan_ms": round(vs["ms_scan"], 3), "records": vs["n_records"], "snv": vs["n_snv"],
                                        "k_vc_depth_gbs": round(16.0 * G / max(vs["ms_depth"], 1e-6) / 1e6, 1),
                                        "k_vc_scan_gbs": round(20.29 * G / max(vs["ms_scan"], 1e-6) / 1e6, 1)}
        except Exception as e:  # never lose the bench line to the optional section
            vcf = {"error": str(e)[:300]}

    if rank == 0:
        total_reads = reads_per_step * args.steps * world
        # dominant kernel: k_seed.  Algorithmic bytes per launch = SURVEY.md §8d's seeding term,
        # 64 * 1.107 * E + rlen per read, with E (FM extension steps) counted by the kernel.  The
        # blocks the kernel really fetches are fewer (k-mer jump table) and reported beside it.
        seed_bytes = 64.0 * 1.107 * d["fm_ext_steps"] + float(d["reads"]) * args.rlen
        prof = pmc_profile(args) or {}
        seed_ms = d["ms_seed"] / max(args.steps, 1)
        achieved = seed_bytes / args.steps / (seed_ms * 1e-3) / 1e9 if seed_ms > 0 else 0.0
        out = {
            "metric": "reads/sec (150 bp PE vs GRCh38) at 1/2/4/8 MI355X; SAM CIGAR bit-exact",
            "value": round(total_reads / dt, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64/int8", "data": "synthetic",
            "config": {"workload": f"synthetic GRCh38-sized genome, {args.genome_mbp:.0f} Mbp ({args.contigs} contigs, {args.repeats} x4 1-kb repeat families; GRCh38 itself is unavailable offline), "
                                   f"{args.batch_pairs} pairs x {args.rlen} bp PE per step per GPU (sub {args.sub}, ins {args.ins}, del {args.dele} per base), -alg {args.alg}",
                       "reads_per_step_per_gpu": reads_per_step, "full_sa_in_hbm": bool(args.full_sa), "index_build_s": round(t_index, 2),
                       "index_hbm_gb": round(index.hbm_bytes / 1e9, 2)},
            "roofline": {"bound": "hbm", "kernel": "k_seed", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": prof.get("traffic"),
                         "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE + WRITE_SIZE, profiles/round1/summary_3gbp_r1d.json)",
                         "algorithmic_bytes_per_launch": round(seed_bytes / max(args.steps, 1)),
                         "note": "achieved prices SURVEY 8d's seeding bytes (64*1.107*E + rlen per read: the reference's FM walk) at the measured "
                                 "launch time; the kernel reaches the same seeds through a K-mer jump table and direct genome comparison, moves far "
                                 "fewer bytes (traffic) and is bound by per-lane request rate, so frac can exceed 1 — it is a speed-of-light "
                                 "comparison with a perfect HBM-bound walk, not an HBM utilisation",
                         "algorithmic_bytes_per_read": round(seed_bytes / max(d["reads"], 1), 1), "avg_launch_ms": round(seed_ms, 3),
                         # SURVEY 8d's own figure for the whole path: achieved = reads/s x B_read against the HBM peak,
                         # B_read = 64 (1.107 E + 31 H) + 8 H + rlen + rlen/4 with E and H counted by the kernels
                         "path": path_roofline(d, args, total_reads / dt / world),  # per GPU
                         # the FM walk is what SURVEY 8d's roofline is about; by launch time another kernel may be longer
                         "longest_launch": max((("k_seed", "ms_seed"), ("k_cluster", "ms_cluster"), ("k_build", "ms_build"), ("k_finish", "ms_finish")),
                                               key=lambda kv: d[kv[1]])[0],
                         "measured": None if not prof else {
                             "hbm_gbs": round(prof["traffic"] / (seed_ms * 1e-3) / 1e9, 1), "l2_requests_per_launch": prof["l2_requests"],
                             "l2_request_rate_g_per_s": round(prof["l2_requests"] / (seed_ms * 1e-3) / 1e9, 1), "waves_waiting_frac": prof["wait_frac"],
                             "reading": "the launch moves less than a tenth of the walk's bytes; waves wait on dependent fetches two thirds of the time"}},
            "per_read": {"fm_ext_steps": round(d["fm_ext_steps"] / max(d["reads"], 1), 2), "fm_blocks": round(d["fm_blocks"] / max(d["reads"], 1), 2),
                         "sa_hits": round(d["sa_hits"] / max(d["reads"], 1), 3), "dp_jobs": round(d["dp_jobs"] / max(d["reads"], 1), 4),
                         "mapped_frac": round(d["mapped"] / max(d["reads"], 1), 4)},
            "stage_ms_per_step": {k[3:]: round(d[k] / args.steps, 3) for k in d if k.startswith("ms_")},
            # measured HBM bytes of one launch (committed PMC passes) over the live stage time: what each big kernel really draws from HBM
            "hbm_utilisation": None if not prof else {
                kern: {"gbs": round(prof["per_kernel"][kern] / (d[ms] / args.steps * 1e-3) / 1e9, 1),
                       "frac_of_peak": round(prof["per_kernel"][kern] / (d[ms] / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)}
                for kern, ms in (("k_pack_reads", "ms_encode"), ("k_seed", "ms_seed"), ("k_cluster", "ms_cluster"), ("k_build", "ms_build"), ("k_finish", "ms_finish"))
                if kern in prof.get("per_kernel", {}) and d[ms] > 0},
            "tier1_pairs": d["tier1_pairs"], "replayed_pairs": d["replayed_pairs"],
        }
        if vcf is not None:
            out["vcf_reduce"] = vcf
        if sample is not None:
            try:
                out["cpu_baseline"] = cpu_baseline(args, index, sample)
            except Exception as e:  # the baseline must never take the bench line down
                out["cpu_baseline"] = {"error": str(e)[:200]}
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
