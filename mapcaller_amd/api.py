"""ctypes binding of include/mcx.h (libmcx.so) — the only way Python reaches the hot path.

There is no Python or CPU fallback: if libmcx.so is missing, or no GPU is visible when a device
function is called, an exception is raised.  torch tensors are accepted for device buffers only
as raw pointers (torch is plumbing for HBM allocation and torch.distributed).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCX_LIB") or os.path.join(_HERE, "libmcx.so")  # (MCX_LIB: another build of the same library, for experiments)
CIGAR_STRIDE = 32
CIGAR_SLACK = 65536


def cigar_pool_words(n_reads):
    """MCX_CIGAR_POOL_WORDS (include/mcx.h): the capacity of a batch's CIGAR pool."""
    return n_reads * CIGAR_STRIDE + CIGAR_SLACK


def pack_reads(bases, lens=None):
    """What mcx_stream_submit_packed takes, from ASCII reads: ``bases`` a torch uint8 tensor [n_reads, rlen] (any device; ``lens``
    optional int tensor of the reads' lengths).  Returns pinned host tensors (codes int32 [n_reads, row_words], len int32 [n_reads],
    odd int64 [n_odd]) — 2-bit codes sixteen bases to a word, first base on top; every byte that is not one of ACGT listed with its place."""
    import torch
    n, rlen = bases.shape
    row_words = (rlen + 15) // 16
    lut = torch.zeros(256, dtype=torch.int64, device=bases.device)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
    known = torch.zeros(256, dtype=torch.bool, device=bases.device)
    for ch in b"ACGT":
        known[ch] = True
    b64 = bases.long()
    if lens is None:
        lens_t = torch.full((n,), rlen, dtype=torch.int64, device=bases.device)
    else:
        lens_t = lens.to(bases.device).long()
    inside = torch.arange(rlen, device=bases.device)[None, :] < lens_t[:, None]
    codes = lut[b64] * inside
    pad = row_words * 16 - rlen
    if pad:
        codes = torch.nn.functional.pad(codes, (0, pad))
    shifts = (30 - 2 * torch.arange(16, device=bases.device)).long()
    words = (codes.reshape(n, row_words, 16) << shifts).sum(-1)
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)
    bad = (~known[b64]) & inside
    r, pos = torch.nonzero(bad, as_tuple=True)
    odd = (r << 32) | (pos << 8) | b64[r, pos]
    return (words.contiguous().cpu().pin_memory(), lens_t.to(torch.int32).cpu().pin_memory(), odd.cpu().pin_memory() if odd.numel() else torch.zeros(1, dtype=torch.int64).pin_memory(),
            int(odd.numel()), row_words)


# every symbol include/mcx.h declares
SYMBOLS = [
    "mcx_last_error", "mcx_device_count", "mcx_pack_row", "mcx_host_cpus", "mcx_gz_inflate", "mcx_index_load", "mcx_index_build", "mcx_index_from_codes", "mcx_index_save", "mcx_index_free", "mcx_index_trim",
    "mcx_index_genome_size", "mcx_index_n_chr", "mcx_index_chr_name", "mcx_index_chr_len", "mcx_index_hbm_bytes",
    "mcx_opts_default", "mcx_ctx_create", "mcx_ctx_create_fit", "mcx_ctx_free", "mcx_bwt_search_batch", "mcx_extend_batch",
    "mcx_avg_init", "mcx_map_batch_dev", "mcx_map_batch", "mcx_cigar_words", "mcx_map_files", "mcx_map_files_ex", "mcx_file_opts_default",
    "mcx_profile_attach", "mcx_profile_settle", "mcx_profile_finalize", "mcx_profile_sparse", "mcx_planes_alloc", "mcx_planes_free", "mcx_planes_bytes",
    "mcx_vcf_defaults", "mcx_call_variants",
    "mcx_batch_begin", "mcx_batch_sums", "mcx_batch_replay", "mcx_batch_end", "mcx_avg_walk", "mcx_batch_totals", "mcx_batch_check", "mcx_avg_advance", "mcx_exchange_local", "mcx_exchange_local_free",
    "mcx_profile_sparse_shard", "mcx_batch_end_keys", "mcx_batch_accumulate",
    "mcx_stream_submit", "mcx_stream_submit_packed", "mcx_stream_map", "mcx_stream_collect", "mcx_stream_next", "mcx_stream_mapped",
    "mcx_stream_map32", "mcx_stream_mapped32",
]
# include/mcx_comm.h (libmcx_comm.so: the RCCL side, loaded by the native CLI only)
COMM_LIB_PATH = os.path.join(_HERE, "libmcx_comm.so")
COMM_SYMBOLS = ["mcx_comm_init_all", "mcx_comm_unique_id", "mcx_comm_init_rank", "mcx_comm_free", "mcx_comm_rank", "mcx_comm_size",
                "mcx_profile_reduce", "mcx_comm_exchange", "mcx_comm_exchange_free"]


class McxError(RuntimeError):
    pass


class Opts(C.Structure):
    _fields_ = [("alg", C.c_int32), ("max_pos_diff", C.c_int32), ("max_mismatch_rate", C.c_float),
                ("max_read_len", C.c_int32), ("max_batch_reads", C.c_int64)]


class Fit(C.Structure):  # mcx_fit
    _fields_ = [("pair_records_trimmed", C.c_int32), ("batch_halvings", C.c_int32), ("single_detail_set", C.c_int32), ("pad", C.c_int32),
                ("max_batch_reads", C.c_int64), ("hbm_free_bytes", C.c_int64), ("hbm_taken_bytes", C.c_int64)]


class Aln(C.Structure):
    _fields_ = [("pos", C.c_int64), ("mate_pos", C.c_int64), ("chr", C.c_int32), ("flag", C.c_int32),
                ("mapq", C.c_int32), ("tlen", C.c_int32), ("nm", C.c_int32), ("as_", C.c_int32), ("xs", C.c_int32),
                ("n_cigar", C.c_int32), ("fwd", C.c_int32), ("has_mate", C.c_int32), ("cigar_off", C.c_int32), ("pad", C.c_int32)]


ALN32_DTYPE = np.dtype([("pos_lo", "<u4"), ("mate_lo", "<u4"), ("pos_hi", "u1"), ("mate_hi", "u1"), ("mapq", "u1"), ("bits", "u1"), ("tlen", "<i4"), ("flag", "<u2"),
                        ("chr", "<u2"), ("nm", "<i2"), ("as", "<i2"), ("xs", "<i2"), ("n_cigar", "<u2"), ("cigar_off", "<u4")])  # mcx_aln32


def aln32_unpack(a32: np.ndarray) -> np.ndarray:
    """mcx_aln_unpack (include/mcx.h) over an array of mcx_aln32 records: the 64-byte form."""
    o = np.zeros(a32.shape, dtype=ALN_DTYPE)
    o["pos"] = a32["pos_lo"].astype(np.int64) | (a32["pos_hi"].astype(np.int64) << 32)
    o["mate_pos"] = a32["mate_lo"].astype(np.int64) | (a32["mate_hi"].astype(np.int64) << 32)
    o["chr"] = np.where(a32["chr"] == 0xFFFF, -1, a32["chr"].astype(np.int32))
    for f in ("flag", "mapq", "tlen", "nm", "as", "xs", "n_cigar", "cigar_off"):
        o[f] = a32[f]
    o["fwd"] = a32["bits"] & 1
    o["has_mate"] = (a32["bits"] >> 1) & 1
    return o


ALN_DTYPE = np.dtype([("pos", "<i8"), ("mate_pos", "<i8"), ("chr", "<i4"), ("flag", "<i4"), ("mapq", "<i4"),
                      ("tlen", "<i4"), ("nm", "<i4"), ("as", "<i4"), ("xs", "<i4"), ("n_cigar", "<i4"),
                      ("fwd", "<i4"), ("has_mate", "<i4"), ("cigar_off", "<i4"), ("pad", "<i4")])


class SparseRec(C.Structure):
    _fields_ = [("pos", C.c_int64), ("type", C.c_uint8), ("len", C.c_uint8), ("seq", C.c_char * 54)]


PLANES = ("A", "C", "G", "T", "multi_hit", "readCount", "F1", "R2", "F2", "R1")


MULTI_PLANE = 4  # the one plane kept in 32 bits


def planes_stride(G: int) -> int:
    return (G + 63) & ~63


def planes_words(G: int) -> int:
    """32-bit words of the ten counter planes of a genome of G positions (mcx_planes_bytes / 4; csrc/mcx_planes.h): with
    stride = G rounded up to 64, multi_hit as u32 [stride], then A C G T readCount F1 R2 F2 R1 as u16 [stride] each."""
    return planes_stride(G) * 11 // 2


def planes_alloc(G: int, device):
    """Zeroed memory for the ten counter planes of a genome of G positions (what profile_attach takes): int32 [planes_words(G)]."""
    import torch
    return torch.zeros(planes_words(G), dtype=torch.int32, device=device)


def planes_parts(planes, G: int):
    """(multi_hit int32 [stride], the nine 16-bit planes int16 [9, stride] in the order A C G T readCount F1 R2 F2 R1): views."""
    import torch
    st = planes_stride(G)
    return planes[:st], planes[st:].view(torch.int16).reshape(9, st)


def planes_view(planes, G: int, lo: int = 0, hi: Optional[int] = None):
    """The positions [lo, hi) of all ten planes as int32 [10, hi - lo] in PLANES order (a copy)."""
    import torch
    hi = G if hi is None else hi
    multi, half = planes_parts(planes, G)
    out = torch.empty((10, hi - lo), dtype=torch.int32, device=planes.device)
    for k in range(10):
        if k == MULTI_PLANE:
            out[k] = multi[lo:hi]
        else:
            out[k] = half[k if k < MULTI_PLANE else k - 1, lo:hi].to(torch.int32) & 0xFFFF
    return out


def planes_from_rows(rows, device=None):
    """The planes' memory for counters given as int [10, G] in PLANES order (tests): the inverse of planes_view."""
    import torch
    G = rows.shape[1]
    planes = planes_alloc(G, device if device is not None else rows.device)
    multi, half = planes_parts(planes, G)
    multi[:G] = rows[MULTI_PLANE].to(torch.int32)
    for k in range(10):
        if k != MULTI_PLANE:
            v = rows[k].to(torch.int32) & 0xFFFF
            half[k if k < MULTI_PLANE else k - 1, :G] = torch.where(v >= 0x8000, v - 0x10000, v).to(torch.int16)
    return planes


class Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("reads", "mapped", "pairs", "pair_dist_sum", "pair_len_sum", "fm_ext_steps", "fm_blocks",
                                         "sa_hits", "dp_jobs", "dp_cells", "tier1_pairs", "replayed_pairs", "halved_selections", "simple_pairs")] + \
               [(n, C.c_double) for n in ("ms_encode", "ms_seed", "ms_sa", "ms_cluster", "ms_rescue", "ms_build",
                                          "ms_dp", "ms_finish", "ms_total", "ms_simple", "ms_order")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)


class Exchange(C.Structure):
    """mcx_exchange: the collective the shards of one run share (all-gather of host bytes)."""
    _fields_ = [("user", C.c_void_p), ("rank", C.c_int32), ("size", C.c_int32), ("allgather", ALLGATHER)]


def dist_exchange(device=None) -> Exchange:
    """An mcx_exchange over torch.distributed (backend "nccl" = RCCL: staged through ``device``;
    "gloo": host tensors).  Keep the returned object alive while the run uses it."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = dist.get_backend() == "nccl"

    def allgather(user, send, recv, nbytes):
        try:
            if nbytes == 0:
                return 0
            mine = torch.frombuffer((C.c_uint8 * nbytes).from_address(send), dtype=torch.uint8).clone()
            if on_gpu:
                mine = mine.to(device)
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            out = torch.cat(parts).cpu().contiguous()
            C.memmove(recv, out.data_ptr(), world * nbytes)
            return 0
        except Exception as e:  # the C side turns this into an error on every shard
            import sys
            print(f"mcx exchange: {e}", file=sys.stderr)
            return -3

    x = Exchange()
    x.user, x.rank, x.size = None, rank, world
    x.allgather = ALLGATHER(allgather)
    return x


class FileOpts(C.Structure):
    """mcx_file_opts: -p, -t, library append, insert-size state across libraries, sharding."""
    _fields_ = [("interleaved_pairs", C.c_int32), ("host_threads", C.c_int32), ("append_sam", C.c_int32), ("reserved0", C.c_int32),
                ("avg_state", C.POINTER(C.c_int64)), ("shard_rank", C.c_int32), ("shard_count", C.c_int32), ("reserved1", C.c_char_p),
                ("exchange", C.POINTER(Exchange))]


class VcfOpts(C.Structure):
    """mcx_vcf_opts: MapCaller's variant-calling switches (reference src/main.cpp:157-187)."""
    _fields_ = [(n, C.c_int32) for n in ("ploidy", "min_allele_depth", "min_cnv", "min_gap", "fragment_size", "filter", "gvcf",
                                         "monomorphic", "somatic", "max_dup", "max_clip")] + \
               [("freq_thr", C.c_float), ("sample_id", C.c_char_p), ("ref_name", C.c_char_p), ("cmdline", C.c_char_p)]


class VcfStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_snv", "n_ins", "n_del", "n_inv", "n_tnl", "n_records")] + \
               [("avg_read_len", C.c_int32), ("fragment_size", C.c_int32)] + \
               [(n, C.c_double) for n in ("ms_depth", "ms_scan", "ms_total")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def lib() -> C.CDLL:
    """Loads libmcx.so (built by ``make -C mapcaller_amd/csrc`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.environ.get("MCX_NO_TORCH"):  # (a host without torch in it — mapcaller_amd/boundary.py — runs on the system's runtime, which libmcx.so links)
        try:  # torch bundles its own HIP runtime: let it load first so that both share one copy
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise McxError(f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')")
    L = C.CDLL(LIB_PATH)
    L.mcx_last_error.restype = C.c_char_p
    L.mcx_index_genome_size.restype = C.c_int64
    L.mcx_index_hbm_bytes.restype = C.c_int64
    L.mcx_index_chr_name.restype = C.c_char_p
    L.mcx_index_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.mcx_index_build.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.mcx_index_from_codes.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_char_p), C.c_int, C.c_int,
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_double)]
    L.mcx_index_save.argtypes = [C.c_void_p, C.c_char_p]
    L.mcx_index_trim.argtypes = [C.c_void_p, C.c_int]
    for f in (L.mcx_index_free, L.mcx_ctx_free):
        f.argtypes = [C.c_void_p]
        f.restype = None
    for f in (L.mcx_index_genome_size, L.mcx_index_n_chr, L.mcx_index_hbm_bytes):
        f.argtypes = [C.c_void_p]
    L.mcx_index_chr_name.argtypes = [C.c_void_p, C.c_int32]
    L.mcx_index_chr_len.argtypes = [C.c_void_p, C.c_int32]
    L.mcx_opts_default.argtypes = [C.POINTER(Opts)]
    L.mcx_opts_default.restype = None
    L.mcx_ctx_create.argtypes = [C.c_void_p, C.POINTER(Opts), C.POINTER(C.c_void_p)]
    L.mcx_ctx_create_fit.argtypes = [C.c_void_p, C.POINTER(Opts), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(Fit)]
    L.mcx_bwt_search_batch.argtypes = [C.c_void_p] + [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 3
    L.mcx_extend_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_uint32] + [C.c_void_p] * 3
    L.mcx_avg_init.argtypes = [C.POINTER(C.c_int64)]
    L.mcx_avg_init.restype = None
    for f in (L.mcx_map_batch_dev, L.mcx_map_batch):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_int64), C.c_void_p,
                      C.c_void_p, C.POINTER(Stats)]
    L.mcx_map_files.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(Stats)]
    L.mcx_map_files_ex.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(FileOpts), C.c_char_p, C.POINTER(Stats)]
    L.mcx_file_opts_default.argtypes = [C.POINTER(FileOpts)]
    L.mcx_file_opts_default.restype = None
    L.mcx_profile_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.mcx_profile_settle.argtypes = [C.c_void_p]
    L.mcx_profile_finalize.argtypes = [C.c_void_p, C.c_void_p]
    L.mcx_profile_sparse.argtypes = [C.c_void_p, C.POINTER(C.POINTER(SparseRec)), C.POINTER(C.c_uint64)]
    L.mcx_profile_sparse_shard.argtypes = [C.c_void_p, C.POINTER(C.POINTER(SparseRec)), C.POINTER(C.c_uint64)]
    L.mcx_batch_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.mcx_batch_sums.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)] + [C.POINTER(C.POINTER(C.c_uint32))] * 3
    L.mcx_batch_replay.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(Stats)]
    L.mcx_batch_end.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.mcx_batch_end_keys.argtypes = [C.c_void_p, C.POINTER(Stats), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]
    L.mcx_batch_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]
    L.mcx_avg_walk.argtypes = [C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.mcx_avg_walk.restype = None
    L.mcx_batch_totals.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.mcx_batch_check.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_uint32), C.POINTER(Stats)]
    L.mcx_avg_advance.argtypes = [C.POINTER(C.c_int64), C.c_int64, C.c_int64, C.c_int64]
    L.mcx_avg_advance.restype = None
    L.mcx_exchange_local.argtypes = [C.c_int32, C.POINTER(Exchange)]
    L.mcx_exchange_local_free.argtypes = [C.POINTER(Exchange)]
    L.mcx_exchange_local_free.restype = None
    L.mcx_stream_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.mcx_stream_submit_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.mcx_stream_map.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.mcx_stream_map32.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.mcx_stream_mapped32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mcx_stream_collect.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mcx_planes_alloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.mcx_planes_free.argtypes = [C.c_void_p]
    L.mcx_planes_free.restype = None
    L.mcx_planes_bytes.argtypes = [C.c_int64]
    L.mcx_planes_bytes.restype = C.c_uint64
    L.mcx_vcf_defaults.argtypes = [C.POINTER(VcfOpts)]
    L.mcx_vcf_defaults.restype = None
    L.mcx_call_variants.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int64,
                                    C.POINTER(VcfOpts), C.c_char_p, C.POINTER(VcfStats)]
    _lib = L
    return L


def _check(rc: int, what: str):
    if rc != 0:
        raise McxError(f"{what} failed ({rc}): {lib().mcx_last_error().decode()}")


def device_count() -> int:
    return int(lib().mcx_device_count())


class Index:
    """FM-index + reference resident in HBM (mcx_index_load; reference src/bwt_index.cpp:150-258)."""

    def __init__(self, prefix: Optional[str], device: int = 0, full_sa: int = 0):
        # full_sa: 0 the sampled suffix array as on disk; 1 (True) every entry + jump table + rank records; 2 + the pair records
        self._h = C.c_void_p()
        self.device = device
        self.build_seconds = None
        if prefix is not None:
            _check(lib().mcx_index_load(prefix.encode(), device, int(full_sa), C.byref(self._h)), "mcx_index_load")

    @classmethod
    def from_codes(cls, d_codes_ptr: int, chr_lens: List[int], chr_names: Optional[List[str]] = None, device: int = 0,
                   full_sa: int = 0) -> "Index":
        """Builds the index on the GPU from a genome already in HBM (codes 0..3, contigs concatenated)."""
        self = cls(None, device)
        n = len(chr_lens)
        lens = (C.c_int32 * n)(*chr_lens)
        names = (C.c_char_p * n)(*[(chr_names[i] if chr_names else f"chr{i + 1}").encode() for i in range(n)])
        secs = C.c_double()
        _check(lib().mcx_index_from_codes(d_codes_ptr, n, lens, names, device, int(full_sa), C.byref(self._h), C.byref(secs)),
               "mcx_index_from_codes")
        self.build_seconds = secs.value
        return self

    def save(self, prefix: str) -> None:
        _check(lib().mcx_index_save(self._h, prefix.encode()), "mcx_index_save")

    def trim(self, full_sa: int = 1) -> None:
        """Gives back what the index holds above the level ``full_sa`` (1: the pair records of full_sa=2).  Mappers made
        before must have been closed."""
        _check(lib().mcx_index_trim(self._h, int(full_sa)), "mcx_index_trim")

    @staticmethod
    def build(fasta: str, prefix: str, device: int = 0) -> None:
        _check(lib().mcx_index_build(fasta.encode(), prefix.encode(), device), "mcx_index_build")

    @property
    def genome_size(self) -> int:
        return int(lib().mcx_index_genome_size(self._h))

    @property
    def hbm_bytes(self) -> int:
        return int(lib().mcx_index_hbm_bytes(self._h))

    @property
    def chromosomes(self) -> List[Tuple[str, int]]:
        L = lib()
        return [(L.mcx_index_chr_name(self._h, i).decode(), int(L.mcx_index_chr_len(self._h, i)))
                for i in range(L.mcx_index_n_chr(self._h))]

    def call_variants(self, d_planes_ptr: int, sparse, pairs: int, pair_dist_sum: int, pair_len_sum: int, vcf_path: str,
                      **switches) -> dict:
        """VariantCalling() over finalized planes (reference src/VariantCalling.cpp:696-740).
        ``sparse``: the records of Mapper.profile_sparse() (of all ranks); ``pairs`` /
        ``pair_dist_sum`` / ``pair_len_sum``: totals of Mapper.stats.  ``switches``: fields of
        mcx_vcf_opts (ploidy, min_allele_depth, min_cnv, min_gap, fragment_size, filter, gvcf,
        monomorphic, somatic, sample_id, ref_name, cmdline)."""
        o = VcfOpts()
        lib().mcx_vcf_defaults(C.byref(o))
        keep = []
        for k, v in switches.items():
            if isinstance(v, str):
                v = v.encode()
                keep.append(v)
            setattr(o, k, v)
        if isinstance(sparse, np.ndarray):  # raw mcx_sparse_rec bytes
            raw = np.ascontiguousarray(sparse, dtype=np.uint8).reshape(-1, 64)
            st = VcfStats()
            _check(lib().mcx_call_variants(self._h, d_planes_ptr, raw.ctypes.data, raw.shape[0], pairs, pair_dist_sum, pair_len_sum, C.byref(o),
                                           vcf_path.encode(), C.byref(st)), "mcx_call_variants")
            return st.as_dict()
        rows = []  # (type, pos, len byte, payload); a string longer than a record continues in 'C' records behind it
        for t, pos, x in sparse:
            if t in "VT":
                rows.append((t, pos, 0, int(x).to_bytes(8, "little", signed=True)))
            else:
                b = x.encode("latin-1")
                rows.append((t, pos, min(len(b), 255), b[:54]))
                for lo in range(54, len(b), 54):
                    rows.append(("C", pos, len(b[lo:lo + 54]), b[lo:lo + 54]))
        recs = (SparseRec * max(len(rows), 1))()
        for i, (t, pos, ln, payload) in enumerate(rows):
            r = recs[i]
            r.pos, r.type, r.len = pos, ord(t), ln
            C.memmove(C.addressof(r) + 10, payload, len(payload))
        st = VcfStats()
        _check(lib().mcx_call_variants(self._h, d_planes_ptr, recs, len(rows), pairs, pair_dist_sum, pair_len_sum, C.byref(o),
                                       vcf_path.encode(), C.byref(st)), "mcx_call_variants")
        return st.as_dict()

    def close(self):
        if self._h:
            lib().mcx_index_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mapper:
    """One mapping context (mcx_ctx): a GPU, its scratch and the reference's tunables."""

    def __init__(self, index: Index, alg: str = "nw", max_read_len: int = 256, max_batch_reads: int = 1 << 20,
                 max_pos_diff: int = 30, max_mismatch_rate: float = 0.05):
        if alg not in ("nw", "ksw2"):
            raise ValueError("alg must be 'nw' or 'ksw2'")
        self.index = index
        o = Opts()
        lib().mcx_opts_default(C.byref(o))
        o.alg = 0 if alg == "nw" else 1
        o.max_read_len = max_read_len
        o.max_batch_reads = max_batch_reads
        o.max_pos_diff = max_pos_diff
        o.max_mismatch_rate = max_mismatch_rate
        self._h = C.c_void_p()
        _check(lib().mcx_ctx_create(index._h, C.byref(o), C.byref(self._h)), "mcx_ctx_create")
        self.avg = (C.c_int64 * 4)()
        lib().mcx_avg_init(self.avg)
        self.stats = Stats()

    @staticmethod
    def fit_plan(index: Index, alg: str = "nw", max_read_len: int = 256, max_batch_reads: int = 1 << 20, with_profile: bool = False, paired: bool = True) -> dict:
        """What mcx_ctx_create_fit makes of this run on this device — everything the run takes allocated at once (context, tier 0's pair records, with_profile:
        planes and bookkeeping buffers), degraded until 4 GB of HBM stay free: pair records trimmed (the INDEX is changed: mcx_index_trim), batch halved —
        then given back.  Returns the mcx_fit record as a dict; the caller sizes its Mapper with ["max_batch_reads"]."""
        o = Opts()
        lib().mcx_opts_default(C.byref(o))
        o.alg = 0 if alg == "nw" else 1
        o.max_read_len, o.max_batch_reads = max_read_len, max_batch_reads
        h, planes, fit = C.c_void_p(), C.c_void_p(), Fit()
        _check(lib().mcx_ctx_create_fit(index._h, C.byref(o), int(with_profile), int(paired), 0, 5, C.byref(h), C.byref(planes) if with_profile else None, C.byref(fit)), "mcx_ctx_create_fit")
        if planes.value:
            lib().mcx_planes_free(planes)
        lib().mcx_ctx_free(h)
        return {k: int(getattr(fit, k)) for k, _ in Fit._fields_ if k != "pad"}

    def reset(self):
        lib().mcx_avg_init(self.avg)
        self.stats = Stats()

    # ---- whole path ---------------------------------------------------------------------
    def map_files(self, fq1: str, fq2: Optional[str], sam: Optional[str], interleaved: bool = False, threads: int = 0,
                  shard: Optional[Tuple[int, int]] = None, exchange: Optional[Exchange] = None, append_sam: bool = False) -> dict:
        """Files in, SAM out (mcx_map_files_ex).  ``interleaved`` = -p, ``threads`` = -t; ``shard`` =
        (rank, count) with ``exchange``: map every count-th batch of the input stream while the shards
        keep one insert-size trajectory and one duplicate-cap order, and write the batches' lines at their
        final place in ``sam`` (the same path on every shard; shard 0 creates it).  The insert-size state
        (self.avg) carries over from call to call like the reference's globals (a new library starts a
        new 200-read chunk); ``append_sam``: a further library of the same run."""
        st = Stats()
        fo = FileOpts()
        lib().mcx_file_opts_default(C.byref(fo))
        fo.interleaved_pairs, fo.host_threads = int(interleaved), threads
        fo.append_sam = int(append_sam)
        if self.avg[3] % 200:
            self.avg[3] += 200 - self.avg[3] % 200
        fo.avg_state = C.cast(self.avg, C.POINTER(C.c_int64))
        if shard and shard[1] > 1:
            if exchange is None:
                raise ValueError("a sharded run needs an exchange (api.dist_exchange())")
            fo.shard_rank, fo.shard_count = int(shard[0]), int(shard[1])
            fo.exchange = C.pointer(exchange)
        _check(lib().mcx_map_files_ex(self._h, fq1.encode(), (fq2 or "").encode() or None, C.byref(fo), (sam or "").encode() or None,
                                      C.byref(st)), "mcx_map_files_ex")
        return st.as_dict()

    def map_batch(self, bases: np.ndarray, off: np.ndarray, paired: bool):
        """Host buffers: bases uint8 ASCII (concatenated), off uint32 [n+1].  Returns (aln, cigars): the records and, per
        read, its CIGAR words (aln[r].n_cigar of them, taken out of the batch's pool at aln[r].cigar_off)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        n = off.size - 1
        aln = np.zeros(n, dtype=ALN_DTYPE)
        pool = np.zeros(cigar_pool_words(n), dtype=np.uint32)
        _check(lib().mcx_map_batch(self._h, bases.ctypes.data, off.ctypes.data, n, int(paired), self.avg,
                                   aln.ctypes.data, pool.ctypes.data, C.byref(self.stats)), "mcx_map_batch")
        return aln, [pool[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["n_cigar"])] for a in aln]

    @staticmethod
    def stream_outputs(n_reads: int, slots: int = 3, record_bytes: int = 64):
        """Pinned host buffers for map_stream's results: [(records uint8[n_reads * record_bytes], CIGAR pool int32[cigar_pool_words(n_reads)])];
        record_bytes = 32: the records as mcx_aln32 (map_stream_packed(..., out32=True): half the bytes on the way out)."""
        import torch
        return [(torch.empty(n_reads * record_bytes, dtype=torch.uint8).pin_memory(), torch.empty(cigar_pool_words(n_reads), dtype=torch.int32).pin_memory())
                for _ in range(slots)]

    def map_stream(self, host_bases_ptrs, host_off_ptr: int, n_reads: int, paired: bool, outputs=None):
        """A sequence of equally shaped batches from pinned host memory, results to pinned host memory, the copies of one
        batch overlapped with the kernels of its neighbours (mcx_stream_*).  ``outputs``: stream_outputs() (batch i lands in
        slot i % len).  Returns (bytes copied in, bytes copied out)."""
        L = lib()
        k = len(host_bases_ptrs)
        outs = outputs or self.stream_outputs(n_reads, min(k, 3))
        h2d = C.c_uint64()
        d2h = C.c_uint64()
        for i in range(k + 2):  # submit(i); map(i - 1); collect(i - 2)
            if i < k:
                _check(L.mcx_stream_submit(self._h, host_bases_ptrs[i], host_off_ptr, n_reads), "mcx_stream_submit")
            if 1 <= i <= k:
                a, g = outs[(i - 1) % len(outs)]
                _check(L.mcx_stream_map(self._h, int(paired), self.avg, a.data_ptr(), g.data_ptr(), C.byref(self.stats)), "mcx_stream_map")
            if i >= 2:
                _check(L.mcx_stream_collect(self._h, C.byref(h2d), C.byref(d2h)), "mcx_stream_collect")
        return h2d.value, d2h.value

    def map_stream_packed(self, packed, n_reads: int, paired: bool, outputs=None, out32: bool = False):
        """map_stream with the reads as a host parser packs them (pack_reads): ``packed`` = [(codes ptr, row_words, len ptr, odd ptr, n_odd)]
        per batch, all in pinned host memory."""
        L = lib()
        k = len(packed)
        outs = outputs or self.stream_outputs(n_reads, min(k, 3), 32 if out32 else 64)
        h2d = C.c_uint64()
        d2h = C.c_uint64()
        stream_map = L.mcx_stream_map32 if out32 else L.mcx_stream_map
        for i in range(k + 2):
            if i < k:
                codes, row_words, lens, odd, n_odd = packed[i]
                _check(L.mcx_stream_submit_packed(self._h, codes, row_words, lens, n_reads, odd, n_odd), "mcx_stream_submit_packed")
            if 1 <= i <= k:
                a, g = outs[(i - 1) % len(outs)]
                _check(stream_map(self._h, int(paired), self.avg, a.data_ptr(), g.data_ptr(), C.byref(self.stats)), "mcx_stream_map")
            if i >= 2:
                _check(L.mcx_stream_collect(self._h, C.byref(h2d), C.byref(d2h)), "mcx_stream_collect")
        return h2d.value, d2h.value

    def map_batch_dev(self, d_bases_ptr: int, d_off_ptr: int, n_reads: int, paired: bool, d_aln_ptr: int, d_cigar_ptr: int):
        """Device pointers (e.g. torch.Tensor.data_ptr()); results stay in HBM."""
        _check(lib().mcx_map_batch_dev(self._h, d_bases_ptr, d_off_ptr, n_reads, int(paired), self.avg, d_aln_ptr,
                                       d_cigar_ptr, C.byref(self.stats)), "mcx_map_batch_dev")

    # ---- a batch in steps (runs whose batches are mapped by several GPUs) ---------------------
    def batch_begin(self, d_bases_ptr: int, d_off_ptr: int, n_reads: int, paired: bool, est0: int, read_base: int, d_aln_ptr: int, d_cigar_ptr: int):
        _check(lib().mcx_batch_begin(self._h, d_bases_ptr, d_off_ptr, n_reads, int(paired), est0, read_base, d_aln_ptr, d_cigar_ptr,
                                     C.byref(self.stats)), "mcx_batch_begin")

    def batch_sums(self):
        """(pairs, dist, len) per chunk of 100 pairs: uint32 numpy views, valid until the next call on the context."""
        n = C.c_uint32()
        p = [C.POINTER(C.c_uint32)() for _ in range(3)]
        _check(lib().mcx_batch_sums(self._h, C.byref(n), C.byref(p[0]), C.byref(p[1]), C.byref(p[2])), "mcx_batch_sums")
        return tuple(np.ctypeslib.as_array(q, shape=(n.value,)) if n.value else np.zeros(0, np.uint32) for q in p)

    def batch_replay(self, est_chunk: np.ndarray) -> int:
        est_chunk = np.ascontiguousarray(est_chunk, dtype=np.int32)
        n = C.c_uint32()
        _check(lib().mcx_batch_replay(self._h, est_chunk.ctypes.data, C.byref(n), C.byref(self.stats)), "mcx_batch_replay")
        return n.value

    def batch_totals(self) -> Tuple[int, int]:
        """(proper pairs, their summed distance) of the batch as it stands: what the shards of a round tell each other."""
        t = (C.c_int64 * 2)()
        _check(lib().mcx_batch_totals(self._h, t), "mcx_batch_totals")
        return int(t[0]), int(t[1])

    def batch_check(self, state_before, first_of_round: bool) -> int:
        """The chunks of the batch against the trajectory that starts from ``state_before`` = (avgDist at the round's start, pairs and
        distance before this batch's first chunk); the pairs whose estimate moved are re-run.  Returns how many."""
        st = (C.c_int64 * 3)(*[int(v) for v in state_before])
        n = C.c_uint32()
        _check(lib().mcx_batch_check(self._h, st, 1 if first_of_round else 0, C.byref(n), C.byref(self.stats)), "mcx_batch_check")
        return n.value

    def batch_end(self):
        _check(lib().mcx_batch_end(self._h, C.byref(self.stats)), "mcx_batch_end")

    # ---- -vcf bookkeeping ---------------------------------------------------------------
    def profile_attach(self, d_planes_ptr: int, max_dup: int = 5, max_clip: int = 5) -> None:
        """d_planes: zeroed device memory of planes_words(GenomeSize) 32-bit words (planes_alloc), caller-owned so
        that it can be reduced across GPUs; every later map_batch* call accumulates into it."""
        _check(lib().mcx_profile_attach(self._h, d_planes_ptr, max_dup, max_clip), "mcx_profile_attach")

    def profile_settle(self) -> None:
        """The planes kept as differences while the run is mapped become counts: once, after the last
        batch and before the planes are read or reduced over the ranks (implied by profile_finalize)."""
        _check(lib().mcx_profile_settle(self._h), "mcx_profile_settle")

    def profile_finalize(self, d_planes_ptr: int) -> None:
        _check(lib().mcx_profile_finalize(self._h, d_planes_ptr), "mcx_profile_finalize")

    def profile_sparse_raw(self, shard: bool = False, copy: bool = True) -> np.ndarray:
        """The same records as they are (uint8 [n, 64] copies of mcx_sparse_rec), for all-gathers
        and for Index.call_variants without a Python loop.  ``shard``: the form for a run spread over
        several shards (discordant-pair events 'E' instead of the sites they resolve to).  ``copy=False``:
        a view of the library's own array — valid until the next profile_* call or close()."""
        recs = C.POINTER(SparseRec)()
        n = C.c_uint64()
        f = lib().mcx_profile_sparse_shard if shard else lib().mcx_profile_sparse
        _check(f(self._h, C.byref(recs), C.byref(n)), "mcx_profile_sparse")
        if n.value == 0:
            return np.zeros((0, 64), dtype=np.uint8)
        buf = (C.c_uint8 * (64 * n.value)).from_address(C.addressof(recs.contents))
        view = np.frombuffer(buf, dtype=np.uint8).reshape(n.value, 64)
        return view.copy() if copy else view

    def profile_sparse(self):
        """[(type, pos, seq or dist)]: 'I'/'D'/'B' events and 'V'/'T' discordant-site records."""
        recs = C.POINTER(SparseRec)()
        n = C.c_uint64()
        _check(lib().mcx_profile_sparse(self._h, C.byref(recs), C.byref(n)), "mcx_profile_sparse")
        out = []
        for i in range(n.value):
            r = recs[i]
            t = chr(r.type)
            if t in "VT":
                out.append((t, int(r.pos), int.from_bytes(C.string_at(C.addressof(r) + 10, 8), "little", signed=True)))
            elif t == "C":  # the string of the record before continues
                pt, pp, ps = out[-1]
                out[-1] = (pt, pp, ps + C.string_at(C.addressof(r) + 10, min(r.len, 54)).decode("latin-1"))
            else:
                out.append((t, int(r.pos), C.string_at(C.addressof(r) + 10, min(r.len, 54)).decode("latin-1")))
        return out

    # ---- per-call drop-ins --------------------------------------------------------------
    def bwt_search(self, seqs: List[bytes], starts: List[int]):
        """BWT_Search for many (code string, start) queries. Returns (len, freq, loc[n,50])."""
        n = len(seqs)
        off = np.zeros(n + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(s) for s in seqs])
        buf = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
        st = np.asarray(starts, dtype=np.int32)
        ln = np.zeros(n, dtype=np.int32)
        fr = np.zeros(n, dtype=np.int32)
        loc = np.zeros((n, 50), dtype=np.uint64)
        _check(lib().mcx_bwt_search_batch(self._h, buf.ctypes.data, off.ctypes.data, st.ctypes.data, n, ln.ctypes.data,
                                          fr.ctypes.data, loc.ctypes.data), "mcx_bwt_search_batch")
        return ln, fr, loc

    def extend(self, alg: str, qs: List[bytes], ts: List[bytes]):
        """nw_alignment / ksw2_alignment for many (read fragment, genome fragment) pairs (ASCII).
        Returns (list of op strings of 'M','I','D', scores)."""
        n = len(qs)
        qo = np.zeros(n + 1, dtype=np.uint32)
        to = np.zeros(n + 1, dtype=np.uint32)
        qo[1:] = np.cumsum([len(s) for s in qs])
        to[1:] = np.cumsum([len(s) for s in ts])
        qb = np.frombuffer(b"".join(qs), dtype=np.uint8).copy()
        tb = np.frombuffer(b"".join(ts), dtype=np.uint8).copy()
        ops = np.zeros(int(qo[-1]) + int(to[-1]) + 16, dtype=np.uint8)
        ol = np.zeros(n, dtype=np.int32)
        sc = np.zeros(n, dtype=np.int32)
        _check(lib().mcx_extend_batch(self._h, 0 if alg == "nw" else 1, qb.ctypes.data, qo.ctypes.data, tb.ctypes.data,
                                      to.ctypes.data, n, ops.ctypes.data, ol.ctypes.data, sc.ctypes.data), "mcx_extend_batch")
        out = []
        for i in range(n):
            b = int(qo[i]) + int(to[i])
            out.append(ops[b:b + int(ol[i])].tobytes().decode())
        return out, sc

    def close(self):
        if self._h:
            lib().mcx_ctx_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def avg_advance(state, pairs: int, dist: int, n_chunks: int):
    """The trajectory's state after a round that held ``pairs`` proper pairs at the summed distance ``dist`` (in place: a list of three)."""
    st = (C.c_int64 * 3)(*[int(v) for v in state])
    lib().mcx_avg_advance(st, int(pairs), int(dist), int(n_chunks))
    state[0], state[1], state[2] = int(st[0]), int(st[1]), int(st[2])
    return state


def avg_walk(state, pairs: np.ndarray, dist: np.ndarray, want_est: bool = True):
    """mcx_avg_walk: advances state = [avgDist, pairs, distance] over the chunks; returns the per-chunk EstiDistance."""
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32)
    dist = np.ascontiguousarray(dist, dtype=np.uint32)
    st = (C.c_int64 * 3)(*[int(v) for v in state[:3]])
    est = np.zeros(pairs.size, dtype=np.int32) if want_est else None
    lib().mcx_avg_walk(st, pairs.ctypes.data, dist.ctypes.data, pairs.size, est.ctypes.data if want_est else None)
    state[0], state[1], state[2] = st[0], st[1], st[2]
    return est


def apply_ops(q: str, t: str, ops: str) -> Tuple[str, str]:
    """Gapped strings the reference's nw_alignment / ksw2_alignment would leave in s1, s2."""
    a, b, i, j = [], [], 0, 0
    for o in ops:
        if o == "M":
            a.append(q[i]); b.append(t[j]); i += 1; j += 1
        elif o == "I":
            a.append(q[i]); b.append("-"); i += 1
        else:
            a.append("-"); b.append(t[j]); j += 1
    return "".join(a), "".join(b)


def sparse_to_maps_text(sparse) -> str:
    """The text form oracle/_ref/mcref_tool 'P' and mcxo_map_files_profile write (<out>.maps):
    insert / delete / break-point maps in std::map order, then the inversion and translocation
    site lists sorted by position (stable)."""
    ins, dele, brk = {}, {}, {}
    inv, tnl = [], []
    for t, pos, v in sparse:
        if t == "I":
            ins[(pos, v)] = ins.get((pos, v), 0) + 1
        elif t == "D":
            dele[(pos, v)] = dele.get((pos, v), 0) + 1
        elif t == "B":
            brk[pos] = brk.get(pos, 0) + 1
        elif t == "V":
            inv.append((pos, v))
        else:
            tnl.append((pos, v))
    lines = []
    for name, m in (("I", ins), ("D", dele)):
        for (pos, seq) in sorted(m, key=lambda k: (k[0], k[1].encode("latin-1"))):
            lines.append(f"{name} {pos} {seq} {m[(pos, seq)]}")
    for pos in sorted(brk):
        lines.append(f"B {pos} {brk[pos]}")
    for name, lst in (("V", inv), ("T", tnl)):
        for pos, dist in sorted(lst, key=lambda k: k[0]):
            lines.append(f"{name} {pos} {dist}")
    return "".join(l + "\n" for l in lines)
