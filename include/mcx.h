/* include/mcx.h — C ABI of the MI355X seed-and-extend path (libmcx.so).
 *
 * MapCaller has no plugin/FFI interface: its seam is the set of extern C++ functions of
 * reference src/structure.h:223-292 that the mapping driver calls.  This header is the batch
 * C ABI that replaces them; each entry point names the reference interface it stands in for.
 * Plain pointers and sizes only — no C++ or torch types.  INTEGRATION.md shows the shims with
 * the reference's own signatures (BWT_Search, nw_alignment, ksw2_alignment) built on top.
 *
 * Conventions: functions return 0 on success and a negative mcx_status otherwise;
 * mcx_last_error() gives a message.  A context is bound to one GPU and one host thread.
 * Pointers named d_* are device (HBM) pointers, everything else is host memory.
 */
#ifndef MCX_H
#define MCX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcx_index mcx_index;
typedef struct mcx_ctx mcx_ctx;

enum mcx_status {
    MCX_OK = 0,
    MCX_ERR_IO = -1,        /* index or read file missing / truncated */
    MCX_ERR_ARG = -2,
    MCX_ERR_DEVICE = -3,    /* HIP runtime error (no GPU, out of memory, launch failure) */
    MCX_ERR_CAPACITY = -4,  /* a pair exceeded the hard capacities of the last tier */
    MCX_ERR_UNSUPPORTED = -5
};

const char *mcx_last_error(void);
int mcx_device_count(void);

/* ---- index ------------------------------------------------------------------------------
 * Replaces bwa_idx_load + RestoreReferenceInfo (reference src/bwt_index.cpp:150-258, called
 * from src/main.cpp:350-361): parses <prefix>.bwt/.sa/.pac/.ann/.amb (the byte-compatible BWA
 * files `MapCaller index` writes) and stages them in HBM.  full_sa != 0 additionally expands
 * the 1-in-32 sampled suffix array to every row on the GPU (bwt_sa then is one 8-byte gather) and
 * derives the seeding walk's jump table and rank records; full_sa = 2 adds the pair
 * records — BWT_Search's extension (src/bwt_search.cpp:128-151) two bases per step, 4 bytes per text
 * position (25 GB for a human genome): same searches, fewer fetches. */
#define MCX_INDEX_SAMPLED 0
#define MCX_INDEX_FULL 1
#define MCX_INDEX_PAIRS 2
#define MCX_INDEX_PAIRS_IF_ROOM 3 /* the pair records when the device has room for them; else MCX_INDEX_FULL with one line on stderr */
int mcx_index_load(const char *prefix, int device, int full_sa, mcx_index **out);
/* Gives back what the index holds above the level `full_sa` (MCX_INDEX_FULL: the pair records), e.g. before the
 * alignment profile's planes are attached.  Refused (MCX_ERR_ARG) while a context of the index is alive. */
int mcx_index_trim(mcx_index *, int full_sa);
/* Replaces bwa_idx_build (reference src/BWT_Index/bwtindex.c:77; `MapCaller index ref.fa prefix`,
 * src/main.cpp:199-207): builds BWT/occ/SA on the GPU from a FASTA file and writes the same
 * five files. */
int mcx_index_build(const char *fasta_path, const char *prefix, int device);
/* The same construction for a genome that already sits in HBM (one code 0..3 per byte, contigs
 * concatenated): builds the index in place, no files.  mcx_index_save writes the five files. */
int mcx_index_from_codes(const uint8_t *d_codes, int32_t n_chr, const int32_t *chr_len, const char *const *chr_name,
                         int device, int full_sa, mcx_index **out, double *build_seconds);
int mcx_index_save(const mcx_index *, const char *prefix);
/* Releases the index's HBM at once.  Contexts of the index that are still alive must not map any more; freeing them afterwards is allowed (the index's
 * host object stays until the last of them is gone). */
void mcx_index_free(mcx_index *);
int64_t mcx_index_genome_size(const mcx_index *);
int32_t mcx_index_n_chr(const mcx_index *);
const char *mcx_index_chr_name(const mcx_index *, int32_t i);
int32_t mcx_index_chr_len(const mcx_index *, int32_t i);
int64_t mcx_index_hbm_bytes(const mcx_index *);

/* ---- options: the reference's globals that steer the path (src/main.cpp:159-191) ---------- */
typedef struct mcx_opts {
    int32_t alg;             /* 0 = nw (default, NW_ALG main.cpp:166), 1 = ksw2 (-alg) */
    int32_t max_pos_diff;    /* MaxPosDiff, -indel, default 30 */
    float max_mismatch_rate; /* MaxMisMatchRate, -maxmm, default 0.05 */
    int32_t max_read_len;    /* longest read the context must hold (default 256) */
    int64_t max_batch_reads; /* reads per batch the context is sized for */
} mcx_opts;
void mcx_opts_default(mcx_opts *);

/* A context's HBM: what mcx_ctx_create takes (itemised on stderr with MCX_ALLOC_LOG=1), tier 0's pair records with the first batch, and — only while more
 * than 16 GB of the device stay free after it — two growths on demand, each said on stderr under MCX_ALLOC_LOG / MCX_TIMING: the large tier's pair records
 * when a batch sends it more pairs than one pass holds (3 KB per read of the batch to start with, 2-24 GB; up to 262144 pairs), and the scratch of the
 * 65-256-column DP list after a batch that fills it (4 GB -> at most 12 GB, once).  MCX_NO_TIER1_GROW=1 keeps both as created; a size asked for with
 * MCX_TIER1_GB is kept too.  Results do not depend on either (tests/test_gpu_parity.py: the SAM of a run that grows equals one that does not). */
int mcx_ctx_create(const mcx_index *, const mcx_opts *, mcx_ctx **out);
/* The same with everything the run will take from the device taken at once — the context, tier 0's pair records (otherwise allocated by the first batch),
 * and with_profile != 0: the counter planes (mcx_planes_alloc -> *planes; free with mcx_planes_free) and the bookkeeping's buffers (mcx_profile_attach with
 * max_dup / max_clip) — and, when that leaves less than 4 GB of HBM, degraded in a fixed order until it fits, each step said on stderr: the index gives its
 * pair records back (mcx_index_trim), then max_batch_reads is halved, repeatedly (whole 200-read chunks, down to 128 K reads).  *fit (may be NULL) says
 * what was done; the caller cuts its run into batches of fit->max_batch_reads.  paired: the run's batches are read pairs (half the pair records).
 * Replaces the reference's `new MappingRecord_t[GenomeSize]` (main.cpp:366-370), which has the host's memory to fall back on. */
typedef struct mcx_fit {
    int32_t pair_records_trimmed; /* 1: mcx_index_trim was applied */
    int32_t batch_halvings;       /* how often max_batch_reads was halved */
    int32_t single_detail_set;    /* 1: no second set of detail records — a batch's -vcf bookkeeping runs inside its call */
    int32_t pad;
    int64_t max_batch_reads;      /* what the context was made for */
    int64_t hbm_free_bytes;       /* free HBM with everything allocated */
    int64_t hbm_taken_bytes;      /* what the run's allocations took (context, pair records of tier 0, planes, bookkeeping) */
} mcx_fit;
int mcx_ctx_create_fit(mcx_index *, const mcx_opts *, int with_profile, int paired, int max_dup, int max_clip, mcx_ctx **out, uint32_t **planes, mcx_fit *fit);
void mcx_ctx_free(mcx_ctx *);

/* ---- per-call drop-ins ------------------------------------------------------------------
 * BWT_Search (reference src/bwt_search.cpp:121, declared src/structure.h:279; only caller
 * src/ReadMapping.cpp:138).  seqs: n concatenated code strings (0..4), seq_off[n+1]; each is
 * searched from start[i] to its end.  len/freq: n each; loc: n*50 (freq[i] entries used). */
int mcx_bwt_search_batch(mcx_ctx *, const uint8_t *seqs, const uint32_t *seq_off, const int32_t *start, uint32_t n,
                         int32_t *len, int32_t *freq, uint64_t *loc);
/* nw_alignment / ksw2_alignment (reference src/nw_alignment.cpp:18, src/ksw2_alignment.cpp:250;
 * declared src/structure.h:289,292; called from src/ReadAlignment.cpp:186-187).  q = read
 * fragments, t = genome fragments (ASCII), concatenated with offset arrays of n+1 entries.
 * ops receives, for job i, ops_len[i] column codes 'M' (base/base), 'I' ('-' in the genome
 * string) and 'D' ('-' in the read string) at ops + (q_off[i] + t_off[i]); score[i] is
 * ez.score of ksw_extz2_sse for ksw2 and 2x the final s for nw. */
int mcx_extend_batch(mcx_ctx *, int alg, const uint8_t *q, const uint32_t *q_off, const uint8_t *t,
                     const uint32_t *t_off, uint32_t n, uint8_t *ops, int32_t *ops_len, int32_t *score);

/* ---- the whole path -----------------------------------------------------------------------
 * One record per read: what GeneratePairedSamStream / GenerateSingleSamStream print
 * (reference src/SamReport.cpp:324-488) in the default unique mode. */
typedef struct mcx_aln {
    int64_t pos;       /* POS, 1-based; 0 = unmapped */
    int64_t mate_pos;  /* PNEXT */
    int32_t chr;       /* RNAME index, -1 = '*' */
    int32_t flag;      /* FLAG */
    int32_t mapq;      /* MAPQ */
    int32_t tlen;      /* TLEN */
    int32_t nm, as, xs;/* NM:i AS:i XS:i */
    int32_t n_cigar;   /* CIGAR operations: len << 4 | op, M=0 I=1 D=2 S=4 */
    int32_t fwd;       /* 0: SEQ/QUAL are printed reverse-complemented / reversed */
    int32_t has_mate;  /* RNEXT '=' */
    int32_t cigar_off; /* word offset of the read's first operation in the batch's CIGAR pool */
    int32_t pad;       /* 64-byte records */
} mcx_aln;

/* The same record in 32 bytes, for the way out of HBM (mcx_stream_map32 / mcx_stream_mapped32): at 64 bytes a read the copy of a
 * batch's records outlasts the part of the next batch's step in which the host has nothing to wait for, and every wait behind it takes
 * milliseconds (the device boundary).  Positions below 2^40, at most 65 535 contigs, reads of up to 1000 bases:
 * the path's own limits.  mcx_aln_unpack gives the 64-byte form back. */
typedef struct mcx_aln32 {
    uint32_t pos_lo, mate_lo;   /* POS, PNEXT: low 32 bits */
    uint8_t pos_hi, mate_hi;    /* ... bits 32-39 */
    uint8_t mapq;
    uint8_t bits;               /* 1: fwd, 2: has_mate */
    int32_t tlen;
    uint16_t flag;
    uint16_t chr;               /* 0xFFFF = '*' */
    int16_t nm, as, xs;
    uint16_t n_cigar;
    uint32_t cigar_off;
} mcx_aln32;
static inline void mcx_aln_unpack(const mcx_aln32 *p, mcx_aln *o)
{
    o->pos = (int64_t)p->pos_lo | ((int64_t)p->pos_hi << 32); o->mate_pos = (int64_t)p->mate_lo | ((int64_t)p->mate_hi << 32);
    o->chr = p->chr == 0xFFFFu ? -1 : (int32_t)p->chr; o->flag = p->flag; o->mapq = p->mapq; o->tlen = p->tlen;
    o->nm = p->nm; o->as = p->as; o->xs = p->xs; o->n_cigar = p->n_cigar; o->fwd = p->bits & 1; o->has_mate = (p->bits >> 1) & 1;
    o->cigar_off = (int32_t)p->cigar_off; o->pad = 0;
}

typedef struct mcx_stats {
    int64_t reads, mapped, pairs, pair_dist_sum;
    int64_t pair_len_sum;   /* ReadLengthSum: bases of the reads counted in `pairs` (ReadMapping.cpp:529-530) */
    int64_t fm_ext_steps;   /* E of SURVEY.md §8d: sum of BWT_Search lengths */
    int64_t fm_blocks;      /* index records the extension walk fetched: 16-byte rank records (64-byte .bwt blocks without the full suffix array) */
    int64_t sa_hits;        /* H: suffix-array hits resolved */
    int64_t dp_jobs, dp_cells;
    int64_t tier1_pairs;    /* pairs re-run with the large capacities */
    int64_t replayed_pairs; /* pairs re-run because the avgDist trajectory moved past their validity interval */
    int64_t halved_selections; /* times a selection of pairs was mapped in two halves because a work list ran over */
    int64_t simple_pairs;   /* pairs (reads, single-end) that went from their seeds to their records on the straight-line path (k_simple) */
    double ms_encode /* k_pack_reads */, ms_seed, ms_sa, ms_cluster /* k_cluster alone */, ms_rescue, ms_build, ms_dp, ms_finish, ms_total;
    double ms_simple;       /* the straight-line path: k_simple (collect), k_simple_dp, k_simple (replay) */
    double ms_order;        /* k_order_count + k_order_place */
} mcx_stats;

#define MCX_CIGAR_STRIDE 32 /* CIGAR words per read a batch's pool has room for on average ... */
#define MCX_CIGAR_SLACK 65536 /* ... plus this many: a batch of a few reads with long CIGARs (or re-run pairs, which take fresh words) still fits */
#define MCX_CIGAR_POOL_WORDS(n_reads) ((size_t)(n_reads) * MCX_CIGAR_STRIDE + MCX_CIGAR_SLACK) /* the capacity every d_cigar / cigar argument must have */

/* Replaces the body of ReadMapping() for one batch (reference src/ReadMapping.cpp:416-646):
 * seeding, clustering, pairing, rescue, extension, scoring, flags/MAPQ/CIGAR.  d_bases: ASCII
 * reads in HBM (16-byte aligned, readable up to 32 bytes past the last base: whole 16-byte words
 * are fetched), d_off: n_reads+1 byte offsets (device), paired: mates interleaved.
 * avg_state[4] carries the reference's running insert-size estimate across batches
 * {avgDist, iTotalPairedNum, TotalPairedDistance, reads seen} (ReadMapping.cpp:20-21,:538-539);
 * initialise with mcx_avg_init.  Results (device): d_aln[n_reads] and the batch's CIGAR pool d_cigar
 * (capacity MCX_CIGAR_POOL_WORDS(n_reads) words): the operations of read r are the n_cigar words from
 * d_cigar[d_aln[r].cigar_off] on.  The pool is filled by wavefronts in no particular order (offsets differ
 * from run to run, contents do not); mcx_cigar_words tells how many of its words the last batch took —
 * all a copy to the host has to move. */
void mcx_avg_init(int64_t avg_state[4]);
int mcx_map_batch_dev(mcx_ctx *, const uint8_t *d_bases, const uint32_t *d_off, uint32_t n_reads, int paired,
                      int64_t avg_state[4], mcx_aln *d_aln, uint32_t *d_cigar, mcx_stats *stats);
int mcx_cigar_words(mcx_ctx *, uint32_t *n_words);
/* same with host buffers (pinned staging inside) */
int mcx_map_batch(mcx_ctx *, const uint8_t *bases, const uint32_t *off, uint32_t n_reads, int paired,
                  int64_t avg_state[4], mcx_aln *aln, uint32_t *cigar, mcx_stats *stats);

/* ---- batches from host memory with the copies overlapped with the kernels -------------------------
 * The device boundary of the drop-in: reads arrive in (pinned) host memory, records leave to (pinned)
 * host memory.  Three batches are in flight, each in a slot of HBM of its own: one being copied in on a
 * copy stream, one under the kernels, one being copied out on another copy stream.
 *   mcx_stream_submit   starts the copy of a batch to HBM and returns at once
 *   mcx_stream_map      maps the oldest submitted batch (like mcx_map_batch_dev: the host thread drives
 *                       the kernels), then starts the copy of its records to aln / cigar
 *   mcx_stream_collect  waits until the oldest mapped batch has arrived in host memory
 * so the loop  submit(i+1); map(i); collect(i-1)  keeps PCIe busy in both directions under the kernels.
 * mcx_stream_next / mcx_stream_mapped are mcx_stream_map in two halves for callers that map the batch
 * in steps (mcx_batch_*): the first hands out the batch's place in HBM, the second starts the copy out.
 * bytes_in / bytes_out (may be NULL): bytes moved over the boundary so far. */
int mcx_stream_submit(mcx_ctx *, const uint8_t *bases, const uint32_t *off, uint32_t n_reads);
/* The same with the reads as a host parser packs them — a quarter of the bytes over PCIe: read r has len[r] bases, its 2-bit codes
 * (A 0, C 1, G 2, T 3; sixteen bases to a word, the first in the top bits) are the words codes[r * row_words ..]; every byte of
 * a read that is not one of the upper-case letters ACGT is listed in `odd` as (read << 32 | position << 8 | byte) and has
 * code 0 in the row.  The device restores the ASCII bytes exactly (mcx_stream_next hands out the same d_bases / d_off as after
 * mcx_stream_submit), so nothing downstream can tell the two apart.  row_words >= ceil(longest read / 16).
 * DEFERRED ERROR: the lengths are checked on the device by the kernel that restores the bytes (no wait at the start of the step).  A batch with a read longer
 * than its row / max_read_len, or with more bases than the slot holds, is mapped as n_reads EMPTY reads (every record unmapped) and refused after the fact with
 * MCX_ERR_ARG: by mcx_stream_map / mcx_stream_map32 when they return, and — for the two-half form mcx_stream_next + mcx_map_batch_dev / mcx_batch_* +
 * mcx_stream_mapped / _mapped32 — by the mcx_stream_collect that hands the batch's records over (the slot is released either way; its records are not results). */
int mcx_stream_submit_packed(mcx_ctx *, const uint32_t *codes, uint32_t row_words, const uint32_t *len, uint32_t n_reads, const uint64_t *odd,
                             uint32_t n_odd);
/* One read's row for mcx_stream_submit_packed, the way the file front end makes it (host code, no device involved; sixteen bases at a time where the
 * CPU has BMI2): row[0 .. row_words) written, the read's bytes that are not upper-case ACGT appended to odd[*n_odd ..] as (read << 32 | position << 8 | byte)
 * while *n_odd < odd_cap.  Returns how many such bytes the read holds (more than were room for: the caller's list was too short). */
uint32_t mcx_pack_row(const uint8_t *seq, uint32_t rlen, uint32_t read, uint32_t *row, uint32_t row_words, uint64_t *odd, uint32_t odd_cap, uint32_t *n_odd);
/* The CPUs the host side counts on when it sizes its thread pools: the affinity mask, cut by the cgroup's CPU-time share (cpu.max / cfs_quota) — not the
 * machine's hardware threads; MCX_HOST_CPUS=n overrides. */
uint32_t mcx_host_cpus(void);
/* The file front end's reader for ordinary .gz input by itself (replaces gzGetNextChunk's gzgets, src/GetData.cpp:101-146, for plain gzip streams): the stream is
 * cut into stretches of stretch_bytes compressed bytes (0: 2 MB), block starts are searched for in them, `threads` stretches are inflated side by side and the
 * 32 KB windows between them filled in afterwards (mapcaller_amd/csrc/mcx_pgz.h).  The text goes to out[0 .. cap) as far as it fits (out may be NULL); returns
 * its whole length, -1 when the file cannot be read or is no gzip file, -2 when the stream is damaged (*n_out: the bytes delivered before that). */
int64_t mcx_gz_inflate(const char *path, int threads, uint64_t stretch_bytes, uint8_t *out, uint64_t cap, uint64_t *n_out);
int mcx_stream_map(mcx_ctx *, int paired, int64_t avg_state[4], mcx_aln *aln, uint32_t *cigar, mcx_stats *stats);
int mcx_stream_collect(mcx_ctx *, uint64_t *bytes_in, uint64_t *bytes_out);
int mcx_stream_next(mcx_ctx *, const uint8_t **d_bases, const uint32_t **d_off, uint32_t *n_reads, mcx_aln **d_aln, uint32_t **d_cigar);
int mcx_stream_mapped(mcx_ctx *, mcx_aln *aln, uint32_t *cigar);
/* mcx_stream_map / mcx_stream_mapped with the records leaving HBM in 32 bytes each (packed on the device, half the bytes across PCIe) */
int mcx_stream_map32(mcx_ctx *, int paired, int64_t avg_state[4], mcx_aln32 *aln, uint32_t *cigar, mcx_stats *stats);
int mcx_stream_mapped32(mcx_ctx *, mcx_aln32 *aln, uint32_t *cigar);

/* ---- a batch in steps: runs whose batches are mapped by several GPUs ---------------------------
 * The reference re-estimates the insert size after every 200-read chunk (ReadMapping.cpp:462,
 * :538-539), one trajectory over the whole input stream.  mcx_map_batch* walks it within a batch;
 * when consecutive batches are mapped by different GPUs the trajectory has to be walked over all of
 * them, so the batch is exposed in steps and the caller exchanges the per-chunk sums between them:
 *   mcx_batch_begin   maps every pair with EstiDistance est0 (the best guess at hand)
 *   mcx_batch_sums    per chunk of 100 pairs: proper pairs, their summed distance and read lengths
 *                     (host arrays, valid until the next call on the context)
 *   mcx_batch_replay  est_chunk[k] = the EstiDistance the single-stream run uses for chunk k: pairs
 *                     whose outcome depends on the difference are mapped again; n_redone = how many.
 *                     Repeat sums -> replay until no shard re-ran a pair.
 *   mcx_batch_end     closes the batch (long-CIGAR pool, -vcf bookkeeping with local admission)
 * read_base = reads of the run that precede this batch in input order (orders the discordant-pair
 * events of mcx_profile_sparse across shards; must be even for paired batches). */
int mcx_batch_begin(mcx_ctx *, const uint8_t *d_bases, const uint32_t *d_off, uint32_t n_reads, int paired, int32_t est0,
                    int64_t read_base, mcx_aln *d_aln, uint32_t *d_cigar, mcx_stats *stats);
int mcx_batch_sums(mcx_ctx *, uint32_t *n_chunks, const uint32_t **pairs, const uint32_t **dist, const uint32_t **len);
int mcx_batch_replay(mcx_ctx *, const int32_t *est_chunk, uint32_t *n_redone, mcx_stats *stats);
int mcx_batch_end(mcx_ctx *, mcx_stats *stats);
/* The reference's walk itself (ReadMapping.cpp:538-539): state = {avgDist, iTotalPairedNum,
 * TotalPairedDistance}; fills est_chunk[0..n_chunks) with the EstiDistance each chunk is mapped
 * with and advances the state over the chunks. */
void mcx_avg_walk(int64_t state[3], const uint32_t *pairs, const uint32_t *dist, uint32_t n_chunks, int32_t *est_chunk);
/* The same feedback with nothing but TOTALS between the shards of a run (24 bytes a shard and round instead of its per-chunk
 * sums): the walk has a closed form — the estimate before a chunk is the round's starting one, or, once more than 1000 proper
 * pairs lie before the chunk, the rounded mean distance over everything before it — so a shard needs of the shards before
 * it (in input order) only how many proper pairs they held and at which summed distance.
 *   mcx_batch_totals  {proper pairs, summed distance} of the batch as it stands
 *   mcx_batch_check   state_before = {avgDist at the round's start, pairs and distance before this batch's first chunk (the
 *                     round's starting totals + those of the shards before this one)}; first_of_round: the batch is the
 *                     round's first (its first chunk takes avgDist as it is).  The chunks' estimates are made on the device,
 *                     the pairs whose estimate moved are listed and re-run (as mcx_batch_replay does); *n_redone.
 *   mcx_avg_advance   the state after a round: state += totals; avgDist follows when the round held a chunk at all
 * A round: begin, then { totals -> exchange -> check } until no shard re-ran a pair, then advance and end. */
int mcx_batch_totals(mcx_ctx *, int64_t totals[2]);
int mcx_batch_check(mcx_ctx *, const int64_t state_before[3], int first_of_round, uint32_t *n_redone, mcx_stats *stats);
void mcx_avg_advance(int64_t state[3], int64_t pairs, int64_t dist, int64_t n_chunks);

/* ---- exchange between the shards of one run (one shard = one GPU) --------------------------------
 * Collective: every shard calls allgather the same number of times; `bytes` is the same on every
 * shard; recv receives size * bytes in rank order.  Host memory.  mapcaller-mi355x -gpus N provides
 * one between its host threads, mapcaller_amd/run.py one over torch.distributed. */
typedef struct mcx_exchange {
    void *user;
    int32_t rank, size;
    int (*allgather)(void *user, const void *send, void *recv, uint64_t bytes);
} mcx_exchange;
/* An in-process exchange for `size` host threads of one process (one per GPU): mcx_exchange_local
 * fills size entries of out[] that share one rendezvous; mcx_exchange_local_free releases it. */
int mcx_exchange_local(int32_t size, mcx_exchange *out);
void mcx_exchange_local_free(mcx_exchange *first);

/* ---- -vcf bookkeeping -------------------------------------------------------------------------
 * Replaces UpdateProfile / UpdateMultiHitCount (reference src/AlignmentProfile.cpp:41-271, called
 * under ProfileLock from src/ReadMapping.cpp:562-573) and the discordant-site lists
 * (src/ReadMapping.cpp:486-521).  d_planes: caller-owned, zero-initialised device memory of
 * mcx_planes_bytes(GenomeSize) bytes (mcx_planes_alloc makes it) — MappingRecord_t
 * (src/structure.h:152-163) unpacked into one plane per counter, each in the width it needs, 22 bytes
 * per position: with stride = GenomeSize rounded up to 64, first multi_hit as u32 [stride], then
 * A C G T readCount F1 R2 F2 R1 as u16 [stride] each (mapcaller_amd/csrc/mcx_planes.h says why 16
 * bits are exact for those nine) — so that several GPUs can sum their arrays
 * with one all-reduce.  Once attached, every mcx_map_batch* call adds its reads.  While a run is
 * being mapped the strand and multi_hit planes (and a context-owned plane for exact-seed coverage)
 * hold DIFFERENCES (+1 where a read starts to cover, -1 behind its end): mcx_profile_settle turns
 * them into counts, once, after the run's last batch and BEFORE the planes are read or summed
 * across GPUs; after it the context takes no more reads until the profile is attached again.
 * mcx_profile_finalize applies the reference's field widths (12-bit saturation at 4095, 16-bit
 * wrap, duplicate cap) in place; call it once, after the last batch (and after the reduce).
 * mcx_profile_sparse returns the insert / delete / break-point tallies ('I','D','B': one record
 * per event, to be summed by (pos, seq)) and the inversion / translocation site records ('V','T':
 * pos = gPos, dist in the first 8 bytes of seq).  max_dup = iMaxDuplicate (-dup, default 5),
 * max_clip = MaxClipSize (-maxclip, default 5).
 * On one shard the bookkeeping of a batch — duplicate check, the planes' updates, the tallies — is QUEUED behind the batch
 * when its call returns and runs under the next batch's kernels (a second set of per-read detail records and a copy of the
 * batch's reads in the context: the caller's buffers are the caller's again on return); mcx_profile_settle / _finalize /
 * _sparse*, the next batch's end and mcx_ctx_free wait for it — the planes are not to be read before one of them, as before.
 * Errors of a batch's bookkeeping (a list that ran over) come back from the next of these calls.  Without room in HBM for
 * the second set, or with MCX_NO_PROF_OVERLAP=1, it runs inside the batch's call. */
typedef struct mcx_sparse_rec {
    int64_t pos;
    uint8_t type;
    uint8_t len;   /* 'I','D': length of the whole string (up to 255); a string longer than seq continues in the */
    char seq[54];  /* records that directly follow ('C': len = bytes held).  Keep a list's records in order.      */
} mcx_sparse_rec;
int mcx_profile_attach(mcx_ctx *, uint32_t *d_planes, int max_dup, int max_clip);
int mcx_profile_settle(mcx_ctx *);                       /* idempotent; implied by mcx_profile_finalize on the attached planes */
int mcx_profile_finalize(mcx_ctx *, uint32_t *d_planes);
int mcx_profile_sparse(mcx_ctx *, const mcx_sparse_rec **recs, uint64_t *n);
/* For a run spread over several shards: the same tallies, but the discordant-pair events as they
 * were seen ('E': pos = pair number in input order, len = branch of ReadMapping.cpp:486-521,
 * seq = g1, g2, dist) instead of 'V'/'T' — the reference's second branch reports whatever the
 * previous discordant pair of the *whole stream* left behind (ReadMapping.cpp:418, :499-505), so the
 * events of all shards are replayed in input order by mcx_call_variants, which takes both forms. */
int mcx_profile_sparse_shard(mcx_ctx *, const mcx_sparse_rec **recs, uint64_t *n);
/* Duplicate cap across shards (AlignmentProfile.cpp:76-77 admits the first max_dup uniquely mapped
 * reads per start position in input order).  Between mcx_batch_end_keys (instead of mcx_batch_end)
 * and mcx_batch_accumulate the caller gathers the keys of all shards of the round:
 *   mcx_batch_end_keys    closes the batch; keys (host, pinned) = (start position << 32 | read
 *                         index in the batch) of the reads that reach the duplicate check
 *   mcx_batch_accumulate  all_keys = the keys of every shard of the round with the index replaced
 *                         by (slot * slot_stride + index), slots in input order; own reads are those
 *                         of slot own_slot.  Admission is decided over all of them against the
 *                         readCount plane, which every shard keeps for the whole run (it is the same
 *                         on all shards and is NOT summed by the reduce); then the own reads are
 *                         accumulated. */
int mcx_batch_end_keys(mcx_ctx *, mcx_stats *stats, const uint64_t **keys, uint64_t *n_keys);
int mcx_batch_accumulate(mcx_ctx *, const uint64_t *all_keys, uint64_t n_all, uint32_t slot_stride, uint32_t own_slot);

/* Device storage for the ten planes of one genome (zero-initialised); free with mcx_planes_free.
 * mcx_planes_bytes: its size (22 bytes per position of the genome rounded up to 64 positions). */
int mcx_planes_alloc(const mcx_index *, uint32_t **d_planes);
void mcx_planes_free(uint32_t *d_planes);
uint64_t mcx_planes_bytes(int64_t genome_size);

/* ---- variant calling ---------------------------------------------------------------------------
 * Replaces VariantCalling() (reference src/VariantCalling.cpp:696-740; called from src/main.cpp:379
 * after Mapping()): block read depth (CalBlockReadDepth :105-121), the per-position scan for
 * SNVs / unmapped gaps / duplicated regions / gVCF and monomorphic records (IdentifyVariants
 * :549-680), indel calls from the insert / delete tallies (GetAreaIndFrequency :63-94), break-point
 * candidates and <INV>/<TNL> calls (:173-340), filters and the VCF text (:404-500).
 * The dense work runs on the GPU over the finalized planes; the sparse tallies are host maps.
 * d_planes: the array given to mcx_profile_attach, after mcx_profile_finalize (and, on several
 * GPUs, after the all-reduce).  recs: every record mcx_profile_sparse returned (all ranks, any
 * order).  paired_pairs / pair_dist_sum / pair_len_sum: totals of mcx_stats over the run
 * (iTotalPairedNum, TotalPairedDistance, ReadLengthSum; src/ReadMapping.cpp:782-790).
 * Options hold the reference's switches; mcx_vcf_defaults fills src/main.cpp:157-187's values. */
typedef struct mcx_vcf_opts {
    int32_t ploidy, min_allele_depth, min_cnv, min_gap, fragment_size; /* -ploidy -ad -min_cnv -min_gap -size */
    int32_t filter, gvcf, monomorphic, somatic;                        /* -filter -gvcf -monomorphic -somatic */
    int32_t max_dup, max_clip;                                         /* -dup -maxclip (used by mcx_profile_attach) */
    float freq_thr;                                                    /* FrequencyThr = 0.2 (no switch) */
    const char *sample_id, *ref_name, *cmdline;                        /* -id; ##reference= and ##command_line= texts */
} mcx_vcf_opts;
typedef struct mcx_vcf_stats {
    int64_t n_snv, n_ins, n_del, n_inv, n_tnl, n_records; /* VarNumVec (VariantCalling.cpp:730) + VCF body lines */
    int32_t avg_read_len, fragment_size;                  /* avgReadLength, FragmentSize used for the break points */
    double ms_depth, ms_scan, ms_total;                   /* k_vc_depth, k_vc_scan (device), whole call (wall) */
} mcx_vcf_stats;
void mcx_vcf_defaults(mcx_vcf_opts *);
int mcx_call_variants(const mcx_index *, const uint32_t *d_planes, const mcx_sparse_rec *recs, uint64_t n_recs,
                      int64_t paired_pairs, int64_t pair_dist_sum, int64_t pair_len_sum,
                      const mcx_vcf_opts *, const char *vcf_path, mcx_vcf_stats *stats);

/* ---- files: MapCaller -i <prefix> -f A [-f2 B] -alg nw|ksw2 -sam out (src/main.cpp:212-321) */
int mcx_map_files(mcx_ctx *, const char *fq1, const char *fq2, const char *sam_path, mcx_stats *stats);
/* The same with the remaining switches of the reference's file loop (src/ReadMapping.cpp:689-760):
 * interleaved_pairs = -p (one file holds both mates alternately), host_threads = -t (parser /
 * formatter threads on the host; 0 = pick), append_sam: a further library of the same run (no
 * header, append), avg_state: carries the insert-size estimate across libraries (NULL = fresh).
 * Sharding over several GPUs (one shard each — threads of one process or processes): the input stream
 * is cut into batches of the context's max_batch_reads, batch k belongs to shard k % shard_count; a
 * shard parses (plain FASTQ: touches) and maps only its batches, and writes their lines at their final
 * place in sam_path — the same path on every shard: shard 0 creates the file and writes the header,
 * the shards learn where their batches' text goes from one another (mcx_file_opts.exchange).  avg_state
 * must be the same on every shard when the run starts; it is again when it ends (avg_state[3] = reads
 * of the whole input). */
typedef struct mcx_file_opts {
    int32_t interleaved_pairs, host_threads, append_sam, reserved0;
    int64_t *avg_state; /* int64_t[4], see mcx_avg_init */
    int32_t shard_rank, shard_count; /* 0, 0: the whole input */
    const char *reserved1;
    const mcx_exchange *exchange;    /* required when shard_count > 1: the shards walk ONE insert-size trajectory and
                                        decide the duplicate cap over ONE input order, so that SAM and profile equal the
                                        single-stream run's (rank/size must equal shard_rank/shard_count) */
} mcx_file_opts;
void mcx_file_opts_default(mcx_file_opts *);
int mcx_map_files_ex(mcx_ctx *, const char *fq1, const char *fq2, const mcx_file_opts *, const char *sam_path, mcx_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
