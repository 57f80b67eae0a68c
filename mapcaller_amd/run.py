#!/usr/bin/env python3
"""MapCaller's command line on one or several MI355X of a node.

    python -m mapcaller_amd.run -i idx -f r1.fq [-f2 r2.fq] [-alg nw|ksw2] [-sam out.sam] [-vcf out.vcf | -no_vcf] ...
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 -m mapcaller_amd.run -i idx -f r1.fq -f2 r2.fq -sam out.sam -vcf out.vcf

One process per GPU.  The input stream is cut into batches that are dealt to the ranks in turn
(reads shard embarrassingly): every rank walks the files, maps its own batches against
its own replica of the index and writes its part of the SAM; rank 0 puts the parts back into input
order.  While they map, the ranks exchange a few KB per round (api.dist_exchange) so that the run
follows the ONE insert-size trajectory and the ONE duplicate-cap order of the input stream: SAM and
VCF equal the single-stream run's.  With -vcf on, the bulk exchange of the run follows
(mapcaller_amd/dist.py): the per-position counter planes are summed onto rank 0 with an RCCL reduce,
the sparse tallies and the run totals gathered, and rank 0 calls the variants (mcx_call_variants).
(The C++ product does the same without Python: mapcaller-mi355x -gpus N.)

Host-side glue only: parsing, mapping, SAM text and variant calling all happen behind the C ABI.
"""
import argparse
import os
import shutil
import sys

import torch

from . import api, dist as mdist


def parse(argv):
    ap = argparse.ArgumentParser(prog="mapcaller_amd.run", add_help=True, prefix_chars="-", allow_abbrev=False)
    ap.add_argument("-i", dest="index", required=True, help="BWT index prefix")
    ap.add_argument("-f", dest="f1", nargs="+", required=True, help="files with #1 mates reads")
    ap.add_argument("-f2", dest="f2", nargs="*", default=[], help="files with #2 mates reads")
    ap.add_argument("-p", "-pair", dest="interleaved", action="store_true", help="paired-end reads are interlaced in the same file")
    ap.add_argument("-alg", default="nw", choices=["nw", "ksw2"])
    ap.add_argument("-sam", default=None)
    ap.add_argument("-vcf", default="output.vcf")
    ap.add_argument("-no_vcf", action="store_true")
    ap.add_argument("-t", dest="threads", type=int, default=0, help="host threads per process for parsing / SAM text")
    ap.add_argument("-batch", type=int, default=1 << 20, help="reads per batch (the unit dealt to the ranks)")
    ap.add_argument("-maxlen", type=int, default=0, help="longest read the contexts are sized for [sampled from the first reads, 256..1000]")
    for name, kw in (("-gvcf", {}), ("-monomorphic", {}), ("-filter", {}), ("-somatic", {})):
        ap.add_argument(name, action="store_true", **kw)
    ap.add_argument("-ploidy", type=int, default=2)
    ap.add_argument("-size", type=int, default=500)
    ap.add_argument("-ad", type=int, default=5)
    ap.add_argument("-dup", type=int, default=5)
    ap.add_argument("-maxclip", type=int, default=5)
    ap.add_argument("-min_cnv", type=int, default=50)
    ap.add_argument("-min_gap", type=int, default=50)
    ap.add_argument("-id", dest="sample", default="unknown")
    ap.add_argument("-backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    a = ap.parse_args(argv)
    if a.f2 and len(a.f2) != len(a.f1):
        ap.error("Paired-end reads input numbers do not match!")
    return a


def sample_read_length(path, lines=40000):
    """Longest sequence line among the first records of a read file (sizes the contexts)."""
    import gzip
    opener = gzip.open if path.endswith(".gz") else open
    longest, fastq = 0, False
    with opener(path, "rb") as fh:
        for n, line in enumerate(fh):
            if n >= lines:
                break
            if n == 0:
                fastq = line.startswith(b"@")
            if (n % 4 == 1) if fastq else (not line.startswith(b">")):
                longest = max(longest, len(line.rstrip(b"\r\n")))
    return longest


def main(argv=None):
    a = parse(sys.argv[1:] if argv is None else argv)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        sys.exit("mapcaller_amd.run needs a GPU: the hot path has no CPU fallback")
    device = local % n_dev
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    td = None
    if world > 1:
        import torch.distributed as td_
        td = td_
        backend = a.backend or "nccl"
        td.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    if a.maxlen <= 0:
        a.maxlen = min(1000, max(256, (max(sample_read_length(f) for f in a.f1 + a.f2) + 63) // 64 * 64))
    index = api.Index(a.index, device=device, full_sa=True)
    mapper = api.Mapper(index, alg=a.alg, max_read_len=a.maxlen, max_batch_reads=max(200, a.batch // 200 * 200))
    want_vcf = not a.no_vcf
    planes = None
    if want_vcf:
        planes = api.planes_alloc(index.genome_size, dev)
        mapper.profile_attach(planes.data_ptr(), max_dup=a.dup, max_clip=a.maxclip)
    totals = {"reads": 0, "mapped": 0, "pairs": 0, "pair_dist_sum": 0, "pair_len_sum": 0}
    link = api.dist_exchange(dev) if world > 1 else None
    for k, f1 in enumerate(a.f1):  # libraries one after the other, like the reference: one SAM stream, one insert-size state
        f2 = a.f2[k] if a.f2 else None
        # (the ranks write into the one SAM file, every batch's lines at their final place: mcx_map_files_ex)
        st = mapper.map_files(f1, f2, a.sam or None, interleaved=a.interleaved, threads=a.threads,
                              shard=(rank, world) if world > 1 else None, exchange=link, append_sam=k > 0)
        for key in totals:
            totals[key] += st[key]
        if td:
            td.barrier()
    tot = mdist.sum_over_ranks([totals[k] for k in ("reads", "mapped", "pairs", "pair_dist_sum", "pair_len_sum")], dev)
    if want_vcf:
        mapper.profile_settle()  # differences -> counts, before the planes are summed
        planes, sparse = mdist.reduce_profile(planes, mapper.profile_sparse_raw(shard=world > 1), index.genome_size, mapper=mapper)
        if rank == 0:
            mapper.profile_finalize(planes.data_ptr())
            vs = index.call_variants(planes.data_ptr(), sparse, tot[2], tot[3], tot[4], a.vcf, ploidy=a.ploidy, min_allele_depth=a.ad,
                                     min_cnv=a.min_cnv, min_gap=a.min_gap, fragment_size=a.size, filter=int(a.filter), gvcf=int(a.gvcf),
                                     monomorphic=int(a.monomorphic), somatic=int(a.somatic), sample_id=a.sample, ref_name=a.index,
                                     cmdline=" ".join(["mapcaller_amd.run"] + (sys.argv[1:] if argv is None else list(argv))))
            print(f"\t{vs['n_snv']}(snp); {vs['n_ins']}(ins); {vs['n_del']}(del); {vs['n_tnl'] >> 1}(trans); {vs['n_inv'] >> 1}(inversion)", file=sys.stderr)
    if rank == 0:
        kind = "paired-end" if (a.f2 or a.interleaved) else "single-end"
        print(f"All the {tot[0]} {kind} reads have been processed on {world} GPU(s).\n{tot[1]:12d} reads are mapped properly.\n"
              f"{2 * tot[2]:12d} reads are mapped in pairs.", file=sys.stderr)
    mapper.close()
    index.close()
    if td:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
