// mapcaller_amd/csrc/mcx_types.h — plain-old-data shared by the HIP kernels, the C ABI and the
// host-side SAM writer.  No STL, no torch: everything here is laid out for HBM.
//
// Design: a batch of read pairs owns a fixed-capacity *pair state* record in
// HBM.  Every stage (seeding, SA resolution, clustering/pairing, DP, scoring/CIGAR) reads and
// writes only its own pair's record, so there are no atomics on the data path and results do
// not depend on scheduling.  A pair that exceeds any capacity is flagged and re-run in the next
// tier, whose capacities are hard upper bounds for the read length.
#ifndef MCX_TYPES_H
#define MCX_TYPES_H
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MCX_HD __host__ __device__
#define MCX_UNROLL _Pragma("unroll")
#else
#define MCX_HD
#define MCX_UNROLL
#endif

namespace mcx {

// constants of the reference path (reference src/structure.h:20-25, src/bwt_search.cpp:3)
enum : int {
    kMinSeedLength = 16,
    kReadChunkSize = 200,
    kOccThr = 50,
    kKmerSize = 8,
    kMinAlnBlockSize = 5,
    kMinTranslocationSize = 1000
};

// FM-index + reference, resident in HBM.  bwt/sa keep the on-disk layout of the reference
// (src/BWT_Index/bwtindex.c:53-75): 64-byte blocks of {4 x u64 occ, 8 x u32 = 128 bases}.
struct IndexView {
    const uint32_t *bwt;
    const uint64_t *sa;      // sampled every sa_intv (sa[0] = ~0)
    const uint64_t *sa_full; // optional: every suffix-array entry (288 GB HBM makes room); may be null
    const uint8_t *pac;      // forward genome, 2 bit/base, MSB first: the genome's .pac bytes
    const int64_t *end_pos;  // sorted chromosome end positions in [0,2G): PosChrIdMap keys
    const int32_t *end_chr;  // chromosome id of each end
    const int64_t *chr_fwd;  // FowardLocation per chromosome
    const uint32_t *ktab;    // optional: bi-interval of every ktab_k-mer (32 B each: x0, x1, x2, pad), x2 = 0 if absent
    int32_t ktab_k;
    // optional (derived at load, mcx_fm.h RankChunk): per base b and 32 BWT symbols one 16-byte record {which symbols equal b, which
    // are greater, how many of either came before}: an extension step of the seeding walk is then two 16-byte fetches, not eight
    const void *rank;        // [4][rank_chunks] records; null: the walk counts in the .bwt blocks
    uint64_t rank_chunks;
    uint64_t rank_cross[8];  // [b]: first chunk whose "equal" count before it is >= 2^32 (the records keep 32 bits); [4 + b]: same for "greater"
    // optional (mcx_fm.h PairSlot): per 32 BWT symbols one 128-byte record over the sixteen PAIRS of bases that precede a suffix, so
    // that the walk extends by two bases with one 16-byte fetch per end of the interval
    const void *rank2;       // [rank_chunks] records; null: one base per step
    const uint64_t *rank2_c2; // [16]: the first row of the suffixes that begin with the pair, minus one (fm_pair_first)
    uint64_t rank2_lone;     // the stored symbol whose suffix has ONE base before it (the suffix at text position 1)
    int32_t rank2_t0;        // that base: the text's first
    uint64_t primary, L2[5], seq_len;
    int64_t G, G2;
    int32_t n_ends, n_chr, sa_intv;
};

struct Params {
    int32_t max_pos_diff;  // -indel, main.cpp:179
    float max_mm_rate;     // -maxmm, main.cpp:186
    int32_t use_nw;        // -alg nw|ksw2
    int32_t paired;        // mates interleaved (2p, 2p+1)
};

// per-tier capacities of one pair-state record
struct Caps {
    int32_t hit_cap;   // seed hits per read (incl. rescue seeds)
    int32_t cand_cap;  // candidates per read
    int32_t hit_seed;  // of hit_cap, what seeding may fill (the rest is room for mate rescue's seeds: a pair that runs over
    int32_t cand_seed; // while clustering goes to the large tier beside the pass; one that runs over in the rescue has to wait for its end)
    int32_t frag_cap;  // fragments per pair (all live candidates of both reads)
    int32_t ops_cap;   // DP op bytes per pair
    int32_t job_cap;   // DP jobs per pair
    int32_t cig_cap;   // CIGAR ops per read
    int32_t kmer_cap;  // rescue window length
};

struct alignas(16) Hit { // one seed occurrence: FragPair_t with bSimple (structure.h:113-123)
    int64_t gPos;     // K1 stores the BWT row here, K2 overwrites it with the text position
    int32_t rPos;
    int32_t len;
};

struct alignas(16) Cand { // AlnCan_t (structure.h:125-133) as ranges into the pair state; two 16-byte groups
    int64_t pd0;      // FragPairVec[0].PosDiff while sorted by PosDiff
    int32_t score;
    int16_t mate;     // PairedAlnCanIdx (-1: none)
    int16_t first;    // first seed (index into the read's hit array)
    int16_t count;    // number of seeds
    int16_t frag_off; // fragments after extension set-up
    int16_t n_frags;
    int16_t flag;     // SamFlag
    int8_t fwd;       // orientation
    uint8_t in_pool;  // the candidate's seeds lie in the pass's seed pool (mate rescue's additions), pool_off on, not among the read's hits
    int16_t pad;
    int32_t pool_off;
};
static_assert(sizeof(Cand) == 32, "Cand is two 16-byte records");

enum FragKind : uint8_t {
    kSimple = 0,  // exact seed
    kPlain = 1,   // gap fragment, rLen == gLen, compared base by base (no DP)
    kIns = 2,     // gLen == 0: read bases against '-'
    kDel = 3,     // rLen == 0: '-' against genome bases
    kDp = 4,      // gapped extension result in the ops pool
    kEmpty = 5    // end fragment dropped by the quality gate
};

// One fragment of a candidate's alignment in 16 bytes — one memory instruction moves it.  The
// per-pair kernels are bound by the number of such instructions, so the fields are bit-fields sized
// for the hard limits of the path: positions below 2^40, reads up to 1000 bases, gap fragments up to
// 2048 x 1024, an ops pool of at most 96 KB.  Signed: lengths are decremented below zero and clamped
// (RemoveOverlaps, ReadAlignment.cpp:38-65).
struct alignas(16) Frag {
    int64_t gPos : 41;
    int64_t rPos : 11;
    int64_t gLen : 12;
    int64_t rLen : 13;
    int64_t ops_off : 18;  // kDp: offset into the pair's ops pool (columns, 'M' 'I' 'D')
    int64_t ops_len : 13;  // current number of alignment columns (after end trimming)
    uint64_t kind : 3;
    uint64_t meta : 14;    // kDp: where the DP kernel left the fragment's DpSummary (ops pool offset / 8 + 1); 0: none, walk the columns.  kPlain: mismatches + 1 as stage_build counted them (0: count again)
};
static_assert(sizeof(Frag) == 16, "Frag is one 16-byte record");

// What the finish stage needs to know about a DP fragment's column string, left in front of the string's area by the lane
// that traced it back (it has both sequences in LDS and visits every column anyway): the finish stage then neither walks
// the columns (three dependent fetches per column, in a loop as long as the wave's longest fragment) nor counts CIGAR runs.
constexpr int kDpRle = 8;    // runs kept; a string with more is walked (n_rle = 0xFFFF)
constexpr int kDpSum = 64;   // bytes reserved in the ops pool in front of every DP problem's columns
struct alignas(8) DpSummary {
    uint32_t cols_off;                    // the untrimmed string: offset in the ops pool
    uint16_t cols_len;
    uint16_t n, mis;                      // 'M' columns; those whose bases differ (FindMisMatchNumber / CheckLocalAlignmentQuality)
    uint16_t switches;                    // runs of one kind of column in the whole string
    uint16_t lead_d, lead_i, lead_runs;   // gap columns and their runs before the first 'M' (RemoveHeadingGaps)
    uint16_t tail_d, tail_i, tail_runs;   // ... after the last 'M' (RemoveTailingGaps)
    uint16_t n_rle;                       // runs in rle[kDpRle - n_rle ..), in column order
    uint32_t rle[kDpRle];                 // (length << 4) | op, op codes as in BAM: M = 0, I = 1, D = 2
};
static_assert(sizeof(DpSummary) <= kDpSum, "the summary fits the room reserved for it");

struct DpJob {        // one ksw2/nw problem
    uint32_t pair;
    uint16_t slot;    // read 0/1 of the pair
    uint16_t rev;     // fragment lies on the reverse strand: both strings reversed
    int32_t rPos, rLen;
    int64_t gPos;
    int32_t gLen;
    int32_t ops_off;  // into the pair's ops pool; capacity rLen + gLen
    int32_t frag;     // fragment index in the pair's fragment pool
    int32_t score;    // in: 1 = the read holds a byte that is not ACGT (its 2-bit words are not used: k_dp_lane2 then needs no look at the read's own words); out: ez.score (ksw2) / final s (nw, doubled)
};

struct ReadSum {      // AlnSummary_t (structure.h:135-140); scores count matched bases (at most the read length)
    int16_t best, score, sub;
};

enum PairFlags : uint32_t {
    kOvHits = 1u, kOvCands = 2u, kOvFrags = 4u, kOvOps = 8u, kOvJobs = 16u, kOvCigar = 32u, kOvKmer = 64u,
    kOvDetail = 128u,
    kOvAny = 255u,
    kRescueUsedEst = 256u,
    kDispatched = 512u, // listed for the large tier while clustering: this tier's later stages and k_finish leave it alone
    kAwaitRescue = 1024u // listed for mate rescue while clustering (never cleared: k_build's first launch, which runs beside the rescue, tells by it which pairs are not its own)
};

struct alignas(16) PairHdr { // 64 bytes: four 16-byte groups
    uint32_t flags;
    int32_t n_frags;
    int32_t n_ops;
    int32_t est;            // EstiDistance used
    int32_t est_lo, est_hi; // the pairing decisions hold for every EstiDistance in [lo, hi]
    int32_t pair_dist;
    int16_t n_hits[2];
    int16_t n_cands[2];
    int16_t n_jobs;
    int16_t n_paired;       // return value of CheckPairedAlignmentDistance / AlignmentRescue
    int16_t pair_ok;        // counted in iTotalPairedNum (ReadMapping.cpp:527-531)
    int16_t mapped;         // number of mapped reads in the pair
    ReadSum sum[2];
    int32_t pad[2];
};
static_assert(sizeof(PairHdr) == 64, "PairHdr is four 16-byte records");

// one output record per read (unique mode: the reference prints exactly one line per read)
struct alignas(16) AlnRec { // 64 bytes: four 16-byte stores
    int64_t pos;        // 1-based POS (0 when unmapped)
    int64_t mate_pos;   // PNEXT (0 = none)
    int32_t chr;        // RNAME index (-1 = *)
    int32_t flag;
    int32_t mapq;
    int32_t tlen;
    int32_t nm, as, xs; // NM AS XS
    int32_t n_cigar;    // ops in the cigar pool row of this read
    int32_t fwd;        // SEQ printed as given (1) or reverse-complemented (0)
    int32_t has_mate;   // RNEXT '=' and PNEXT/TLEN valid
    int32_t pad[2];
};

// Alignment detail of one read, written by the finish stage when the -vcf bookkeeping is on:
// what UpdateProfile / UpdateMultiHitCount (AlignmentProfile.cpp:41-271) read from ReadItem_t.
struct DetailHdr {
    int32_t type;        // 0 unmapped, 1 exactly one surviving candidate, 2 several (multi-hit)
    int32_t n_frags;     // type 1: fragments of the candidate; type 2: ranges {gPos = begin, rLen = length}
    int32_t fwd;         // orientation of the candidate
    int32_t n_ops;
    int32_t disc_kind;   // read 0 of a pair: discordant-pair event of ReadMapping.cpp:486-521 (0 none, 1/2 strand mix, 3/4 distant)
    int32_t frag0;       // index of the record's first fragment (0 but for the straight-line path's reverse-strand reads: mcx_simple.h SimpleDetail)
    int64_t disc_g1, disc_g2, disc_dist;
};

struct DetailLayout {
    int32_t frag_cap, ops_cap;
    int64_t off_ops, stride; // record = DetailHdr, Frag[frag_cap], ops[ops_cap]
};

static inline MCX_HD DetailLayout make_detail_layout(int rlen_max)
{
    DetailLayout d;
    d.frag_cap = 64; d.ops_cap = 2 * rlen_max + 128;
    d.off_ops = (int64_t)sizeof(DetailHdr) + (int64_t)d.frag_cap * sizeof(Frag);
    d.stride = (d.off_ops + d.ops_cap + 63) & ~(int64_t)63;
    return d;
}

// sparse tallies: insert / delete strings, break points (InsertSeqMap, DeleteSeqMap, BreakPointMap)
struct SparseRec {
    int64_t pos;
    uint8_t type;        // 'I', 'D', 'B'
    uint8_t len;
    char seq[54];
};

// per-pair summary the host replays the reference's avgDist feedback from
struct alignas(16) PairOut { // 32 bytes: two 16-byte stores
    uint32_t flags;
    int32_t est, est_lo, est_hi;
    int32_t pair_dist;
    int16_t pair_ok, mapped;
    int32_t pad[2];
};

// byte offsets of the regions of a pair-state record
struct Layout {
    int64_t off_hits, off_cands, off_frags, off_ops, off_jobs, stride;
};

static inline MCX_HD int64_t mcx_align64(int64_t x) { return (x + 63) & ~(int64_t)63; }

static inline MCX_HD Layout make_layout(const Caps &c)
{
    Layout l;
    int64_t o = mcx_align64(sizeof(PairHdr));
    l.off_hits = o;  o += mcx_align64((int64_t)2 * c.hit_cap * sizeof(Hit));
    l.off_cands = o; o += mcx_align64((int64_t)2 * c.cand_cap * sizeof(Cand));
    l.off_frags = o; o += mcx_align64((int64_t)c.frag_cap * sizeof(Frag));
    l.off_jobs = o;  o += mcx_align64((int64_t)c.job_cap * sizeof(int32_t) * 2);
    l.off_ops = o;   o += mcx_align64((int64_t)c.ops_cap);
    l.stride = o;
    return l;
}

struct PairState {
    PairHdr *hdr;
    Hit *hits[2];
    Cand *cands[2];
    Frag *frags;
    uint8_t *ops;
};

static inline MCX_HD PairState pair_state(uint8_t *base, const Layout &l, const Caps &c, int64_t pair)
{
    uint8_t *p = base + pair * l.stride;
    PairState s;
    s.hdr = (PairHdr *)p;
    s.hits[0] = (Hit *)(p + l.off_hits);
    s.hits[1] = s.hits[0] + c.hit_cap;
    s.cands[0] = (Cand *)(p + l.off_cands);
    s.cands[1] = s.cands[0] + c.cand_cap;
    s.frags = (Frag *)(p + l.off_frags);
    s.ops = p + l.off_ops;
    return s;
}

// a batch of reads in HBM: ASCII bases and offsets, as handed over (16-byte aligned base pointer)
struct ReadBatch {
    const uint8_t *bases;   // ASCII
    const uint32_t *off;    // n_reads + 1
    uint32_t n_reads;
};

} // namespace mcx
#endif
