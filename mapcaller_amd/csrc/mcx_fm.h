// mapcaller_amd/csrc/mcx_fm.h — FM-index primitives and the per-read seeding walk.
//
// Replaces, on the device, the reference's bwt_occ4 / bwt_2occ4 / bwt_occ / bwt_invPsi / bwt_sa
// and BWT_Search (reference src/bwt_search.cpp:25-164) plus the greedy seeding loop
// IdentifySimplePairs (src/ReadMapping.cpp:125-158).  Written for one read per lane: every
// extension step is one dependent 64-byte block fetch (two when the interval straddles a
// 128-base block), issued as four 16-byte lane loads, and the per-read loop is flattened into
// a single state machine so that all 64 lanes of a wave issue a block fetch per iteration
// instead of waiting for the longest search of the wave.
#ifndef MCX_FM_H
#define MCX_FM_H
#include "mcx_types.h"

namespace mcx {

struct alignas(16) U4 { uint32_t x, y, z, w; };

struct FmBlock {       // one 64-byte block of the .bwt file
    uint64_t occ[4];   // occurrences of A,C,G,T before the block
    uint32_t w[8];     // 128 bases, 2 bit each, MSB first
};

static inline MCX_HD void fm_load_block(const uint32_t *bwt, uint64_t blk, FmBlock &b)
{
    const U4 *p = (const U4 *)(bwt + (blk << 4));
    U4 a = p[0], c = p[1], d = p[2], e = p[3];
    b.occ[0] = (uint64_t)a.x | ((uint64_t)a.y << 32);
    b.occ[1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
    b.occ[2] = (uint64_t)c.x | ((uint64_t)c.y << 32);
    b.occ[3] = (uint64_t)c.z | ((uint64_t)c.w << 32);
    b.w[0] = d.x; b.w[1] = d.y; b.w[2] = d.z; b.w[3] = d.w;
    b.w[4] = e.x; b.w[5] = e.y; b.w[6] = e.z; b.w[7] = e.w;
}

// The blocks in HBM are a *derived* form of the file's: the file keeps each occurrence count in a u64 of which 34 bits are ever
// used, and a count over the first n symbols of a block walks up to eight words.  fm_derive_block puts, into the unused top 24
// bits of every count, how many of that base the block's first 32, 64 and 96 symbols hold (8 bits each): a count is then the
// stored count + one of those + a popcount over the two words of the 32-symbol stretch that holds symbol n - 1 — a quarter of
// the vector instructions, which is what the seeding kernel is short of.  Same block size, same fetches, same results; the files
// written by mcx_index_save carry the plain counts again (fm_plain_count).
constexpr uint64_t kFmCountMask = (1ull << 40) - 1;
static inline MCX_HD uint64_t fm_plain_count(uint64_t v) { return v & kFmCountMask; }

static inline MCX_HD void fm_derive_block(uint32_t *blk) // the 16 words of one block, in place (idempotent)
{
    uint32_t cnt[4] = {0, 0, 0, 0}, at[3][4];
    for (int j = 0; j < 6; j++) { // the first 96 symbols
        const uint32_t w = blk[8 + j];
        const uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t t = (uint32_t)__popc(hi & lo), c = (uint32_t)__popc(lo), g = (uint32_t)__popc(hi);
#else
        const uint32_t t = (uint32_t)__builtin_popcount(hi & lo), c = (uint32_t)__builtin_popcount(lo), g = (uint32_t)__builtin_popcount(hi);
#endif
        cnt[0] += 16 - c - g + t; cnt[1] += c - t; cnt[2] += g - t; cnt[3] += t;
        if (j & 1) for (int x = 0; x < 4; x++) at[j >> 1][x] = cnt[x];
    }
    for (int x = 0; x < 4; x++) {
        const uint32_t hi = (blk[2 * x + 1] & 0xFFu) | (at[0][x] << 8) | (at[1][x] << 16) | (at[2][x] << 24);
        blk[2 * x + 1] = hi;
    }
}

// the two words of the 32-symbol stretch q of a block, masked to its first m (1..32) symbols: low bits and high bits of the symbols
static inline MCX_HD void fm_stretch(const FmBlock &b, int q, int m, uint32_t &lo0, uint32_t &hi0, uint32_t &lo1, uint32_t &hi1)
{
    // (values first, then the choice: a conditional between array elements is a conditional between addresses, and a block
    //  addressed by a run-time index lives in scratch memory instead of registers)
    const uint32_t a0 = b.w[0], a1 = b.w[1], a2 = b.w[2], a3 = b.w[3], a4 = b.w[4], a5 = b.w[5], a6 = b.w[6], a7 = b.w[7];
    const uint32_t w0 = q == 0 ? a0 : q == 1 ? a2 : q == 2 ? a4 : a6;
    const uint32_t w1 = q == 0 ? a1 : q == 1 ? a3 : q == 2 ? a5 : a7;
    const int m0 = m > 16 ? 16 : m, m1 = m - m0;
    const uint32_t k0 = 0x55555555u & (0xFFFFFFFFu << (32 - 2 * m0)), k1 = m1 ? 0x55555555u & (0xFFFFFFFFu << (32 - 2 * m1)) : 0u;
    lo0 = w0 & k0; hi0 = (w0 >> 1) & k0; lo1 = w1 & k1; hi1 = (w1 >> 1) & k1;
}

// the count stored for base c at the start of stretch q: the block's count + what the stretches before q hold
static inline MCX_HD uint64_t fm_stretch_base(uint64_t occ, int q)
{
    const uint32_t top = (uint32_t)(occ >> 32); // bits 0-7: the count's bits 32-39; 8-15, 16-23, 24-31: symbols 0-31, 0-63, 0-95
    const uint32_t before = q ? (top >> (8 * q)) & 0xFFu : 0u;
    return (occ & kFmCountMask) + before;
}

static inline MCX_HD int fm_popc(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(v);
#else
    return __builtin_popcount(v);
#endif
}

// occurrences of each base among the first n (1..128) symbols of the block, added to occ[]
static inline MCX_HD void fm_count4(const FmBlock &b, int n, uint64_t cnt[4])
{
    const int q = (n - 1) >> 5, m = n - 32 * q;
    uint32_t lo0, hi0, lo1, hi1;
    fm_stretch(b, q, m, lo0, hi0, lo1, hi1);
    const uint32_t t = (uint32_t)(fm_popc(hi0 & lo0) + fm_popc(hi1 & lo1)), ct = (uint32_t)(fm_popc(lo0) + fm_popc(lo1)), gt = (uint32_t)(fm_popc(hi0) + fm_popc(hi1));
    cnt[0] = fm_stretch_base(b.occ[0], q) + ((uint32_t)m - ct - gt + t);
    cnt[1] = fm_stretch_base(b.occ[1], q) + (ct - t);
    cnt[2] = fm_stretch_base(b.occ[2], q) + (gt - t);
    cnt[3] = fm_stretch_base(b.occ[3], q) + t;
}

// occurrences of base c among the first n symbols (bwt_occ, bwt_search.cpp:25-47)
static inline MCX_HD uint64_t fm_count1(const FmBlock &b, int n, int c)
{
    const int q = (n - 1) >> 5, m = n - 32 * q;
    uint32_t lo0, hi0, lo1, hi1;
    fm_stretch(b, q, m, lo0, hi0, lo1, hi1);
    const uint32_t t = (uint32_t)(fm_popc(hi0 & lo0) + fm_popc(hi1 & lo1)), ct = (uint32_t)(fm_popc(lo0) + fm_popc(lo1)), gt = (uint32_t)(fm_popc(hi0) + fm_popc(hi1));
    const uint32_t pc = c == 0 ? (uint32_t)m - ct - gt + t : c == 1 ? ct - t : c == 2 ? gt - t : t;
    const uint64_t o0 = b.occ[0], o1 = b.occ[1], o2 = b.occ[2], o3 = b.occ[3];
    const uint64_t occ = c == 0 ? o0 : c == 1 ? o1 : c == 2 ? o2 : o3;
    return fm_stretch_base(occ, q) + pc;
}

// ---- rank records: the index as the seeding walk wants it ----------------------------------------------------------------
// BWT_Search extends by ONE known base b per step and needs three numbers of bwt_2occ4's eight: occ(k, b), occ(l, b) - occ(k, b)
// and the summed occ(l, c) - occ(k, c) over the bases c > b (bwt_search.cpp:134-151).  For every base b and every 32 BWT symbols
// one 16-byte record holds exactly what those take: a bit per symbol "is b", a bit per symbol "is greater than b", and the number
// of either among all symbols before the record (32 bits each; the one place where such a number passes 2^32 is kept beside the
// tables).  A step is then one 16-byte fetch per end of the interval — two lane fetches where the .bwt block form takes eight —
// and two masked popcounts per end.  The numbers are the same numbers: a derived layout (12 GB for a 3.1 Gbp genome), never a
// different result.  Symbol s of a record is bit 31 - s.
struct alignas(16) RankChunk { uint32_t eq, gt, n_eq, n_gt; };

// even bits (0, 2, .. 30) of t, packed into 16 (bit 2i -> bit i)
static inline MCX_HD uint32_t fm_even16(uint32_t t)
{
    t &= 0x55555555u; t = (t | (t >> 1)) & 0x33333333u; t = (t | (t >> 2)) & 0x0f0f0f0fu; t = (t | (t >> 4)) & 0x00ff00ffu;
    return (t | (t >> 8)) & 0xffffu;
}

// the four records (one per base) of the 32 symbols in words w0, w1 of a block, given the counts of the four bases before them
static inline MCX_HD void fm_rank_records(uint32_t w0, uint32_t w1, const uint64_t before[4], RankChunk out[4], uint64_t n_eq[4], uint64_t n_gt[4])
{
    const uint32_t L = (fm_even16(w0) << 16) | fm_even16(w1), H = (fm_even16(w0 >> 1) << 16) | fm_even16(w1 >> 1); // symbol s at bit 31 - s
    const uint32_t eq[4] = {~H & ~L, ~H & L, H & ~L, H & L}, gt[4] = {H | L, H, H & L, 0u};
    for (int b = 0; b < 4; b++) {
        uint64_t above = 0;
        for (int c = b + 1; c < 4; c++) above += before[c];
        n_eq[b] = before[b]; n_gt[b] = above;
        out[b].eq = eq[b]; out[b].gt = gt[b]; out[b].n_eq = (uint32_t)before[b]; out[b].n_gt = (uint32_t)above;
    }
}

// bwt_2occ4 (bwt_search.cpp:68-99) for k < l, neither equal to (u64)-1 — which is what
// BWT_Search always passes (x1 >= 1).  n_blocks counts the 64-byte blocks touched.
static inline MCX_HD void fm_2occ4(const IndexView &ix, uint64_t k, uint64_t l, uint64_t ck[4], uint64_t cl[4],
                                   int &n_blocks)
{
    k -= (k >= ix.primary);
    l -= (l >= ix.primary);
    FmBlock b;
    fm_load_block(ix.bwt, k >> 7, b);
    fm_count4(b, (int)(k & 127) + 1, ck);
    n_blocks = 1;
    if ((l >> 7) != (k >> 7)) { fm_load_block(ix.bwt, l >> 7, b); n_blocks = 2; }
    fm_count4(b, (int)(l & 127) + 1, cl);
}

// bwt_invPsi (bwt_search.cpp:101-107): one LF step, one block fetch
static inline MCX_HD uint64_t fm_lf(const IndexView &ix, uint64_t k)
{
    if (k == ix.primary) return 0;
    uint64_t x = k - (k > ix.primary);
    FmBlock b;
    fm_load_block(ix.bwt, x >> 7, b);
    int c = (b.w[(x & 127) >> 4] >> ((~x & 15) << 1)) & 3;
    // bwt_occ(k, c): k == seq_len cannot reach here with k != primary only if ... handled:
    if (k == ix.seq_len) return ix.L2[c] + (ix.L2[c + 1] - ix.L2[c]);
    return ix.L2[c] + fm_count1(b, (int)(x & 127) + 1, c);
}

// bwt_sa (bwt_search.cpp:109-119)
static inline MCX_HD uint64_t fm_sa(const IndexView &ix, uint64_t k, int &lf_steps)
{
    if (ix.sa_full) return ix.sa_full[k];
    uint64_t steps = 0, mask = (uint64_t)ix.sa_intv - 1;
    while (k & mask) { ++steps; k = fm_lf(ix, k); }
    lf_steps += (int)steps;
    return steps + ix.sa[k / (uint64_t)ix.sa_intv];
}

// ASCII -> 0..4 (nst_nt4_table, BWT_Index/bntseq.c:40-57)
static inline MCX_HD int nt4_code(uint8_t ch)
{
    // branch-free: fold case, test membership in {A,C,G,T} with a bit mask over (letter - 'A'),
    // and hash the letter: (c >> 1) & 3 gives A0 C1 T2 G3, the xor swaps the last two
    const unsigned c = ch & 0xDFu, d = c - 0x41u;
    const unsigned member = d < 20u ? ((0x80045u >> d) & 1u) : 0u; // bits 0 (A), 2 (C), 6 (G), 19 (T)
    const unsigned h = (c >> 1) & 3u;
    return member ? (int)(h ^ (h >> 1)) : 4;
}

// One read as the kernels see it: the ASCII bases as handed over, and whether it is mate 2 of a
// pair, which the reference reverse-complements in place before anything else
// (ReverseOrientation, tools.cpp:45; EnCodeReadSeq, ReadMapping.cpp:404).  Codes are decoded on
// the fly wherever single bases are needed; the seeding walk works on the 2-bit form k_pack_reads makes.
struct ReadRef {
    const uint8_t *ascii;
    int32_t rlen;
    int32_t flipped;
    // optional: the read's 2-bit words as k_pack_reads leaves them (oriented: mate 2 already reverse-complemented),
    // for reads without N — the fused per-pair kernel keeps them in LDS and never touches the ASCII bases
    const uint32_t *codes = nullptr;
};

static inline MCX_HD int read_code(const ReadRef &r, int i)
{
    if (r.codes) return (int)((r.codes[i >> 4] >> (30 - 2 * (i & 15))) & 3u);
    int c = nt4_code(r.ascii[r.flipped ? r.rlen - 1 - i : i]);
    return (r.flipped && c < 4) ? 3 - c : c;
}

// sequential reader for the seeding walk: keeps the aligned 16-byte chunk that holds the current
// base in registers, so a 150-base read costs ~10 loads instead of 150 byte loads
struct ReadCursor {
    U4 w;
    uintptr_t chunk;
};

static inline MCX_HD int cursor_code(const ReadRef &r, ReadCursor &cur, int i)
{
    const uintptr_t a = (uintptr_t)r.ascii + (uintptr_t)(r.flipped ? r.rlen - 1 - i : i);
    const uintptr_t ch = a & ~(uintptr_t)15;
    if (ch != cur.chunk) { cur.w = *(const U4 *)ch; cur.chunk = ch; }
    const unsigned k = (unsigned)(a & 15);
    const uint32_t word = k < 8 ? (k < 4 ? cur.w.x : cur.w.y) : (k < 12 ? cur.w.z : cur.w.w);
    int c = nt4_code((uint8_t)(word >> ((k & 3) * 8)));
    return (r.flipped && c < 4) ? 3 - c : c;
}

// code of 2G-coordinate p of RefSequence (bwt_index.cpp:196-215): forward strand from the
// packed genome, reverse strand = complement of the mirrored forward base
static inline MCX_HD int ref_code(const IndexView &ix, int64_t p)
{
    bool rev = p >= ix.G;
    int64_t f = rev ? ix.G2 - 1 - p : p;
    int b = (ix.pac[f >> 2] >> ((~f & 3) << 1)) & 3;
    return rev ? 3 - b : b;
}

// sequential reader of RefSequence for the direct-comparison phase of the seeding walk: keeps the
// aligned 16-byte chunk of the 2-bit genome (64 bases) that holds the current base in registers
struct RefCursor {
    U4 w;
    uintptr_t chunk;
};

static inline MCX_HD int ref_cursor_code(const IndexView &ix, RefCursor &cur, int64_t p)
{
    const bool rev = p >= ix.G;
    const int64_t f = rev ? ix.G2 - 1 - p : p;
    const uintptr_t a = (uintptr_t)ix.pac + (uintptr_t)(f >> 2);
    const uintptr_t ch = a & ~(uintptr_t)15;
    if (ch != cur.chunk) { cur.w = *(const U4 *)ch; cur.chunk = ch; }
    const unsigned k = (unsigned)(a & 15);
    const uint32_t word = k < 8 ? (k < 4 ? cur.w.x : cur.w.y) : (k < 12 ? cur.w.z : cur.w.w);
    const int b = (int)((word >> ((k & 3) * 8 + ((~f & 3) << 1))) & 3u);
    return rev ? 3 - b : b;
}

// PosChrIdMap.lower_bound(gPos): first chromosome end >= gPos, -1 past the last one
static inline MCX_HD int end_slot(const IndexView &ix, int64_t gPos)
{
    int lo = 0, hi = ix.n_ends;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (ix.end_pos[mid] < gPos) lo = mid + 1; else hi = mid;
    }
    return lo < ix.n_ends ? lo : -1;
}

// ---- pair records: the walk two bases at a time ---------------------------------------------------------------------------
// Two consecutive extensions of BWT_Search by the bases b1, b2 (bwt_search.cpp:128-151) are one extension over the alphabet of
// PAIRS: the suffix in row r has T[SA[r] - 1] = b1 and T[SA[r] - 2] = b2 before it exactly when it survives both steps, and rows
// that do keep their order.  With the code c1 * 4 + c2 of its two preceding bases per stored BWT symbol,
//     occ2(k, s)  = symbols [0, k] with code s           (what two steps add to the first row of the suffixes beginning b2 b1 ..)
//     gt2(k, s)   = symbols [0, k] with a greater code   (the lower bases' share of the other strand's interval, ok[b].x0 :143-146,
//                                                         both steps' at once: c1 above b1, or c1 = b1 and c2 above b2)
// and the interval after the two steps is {x0 + [primary inside] + gt2(l) - gt2(k), first[s] + 1 + occ2(k), occ2(l) - occ2(k)}.
// One suffix has a single base before it (the one at text position 1: its second step meets the primary row, which the second
// step's "+1" of :143 accounts for): it has no code, counts as greater than the pairs that begin with its base or a lower one,
// and is kept beside the table (rank2_lone, rank2_t0).
// Layout: per 32 symbols 16 slots of 8 bytes {which symbols have a code below j, how many had before the record (low 32 bits)}
// for j = 1..8 in the first 64-byte line and j = 8..15 in the second, so that the two numbers a step needs — below s and below
// s + 1 — are 16 neighbouring bytes of ONE line for every s (below 0 is nothing, below 16 is every symbol with a code).  Counts
// are kept modulo 2^32: the walk takes differences of at most the interval's width, and occ2 itself stays below 2^32 (checked
// when the table is built).  4 bytes per text position.  The numbers are the same numbers as two single steps': never a
// different result; a pair that comes up empty says nothing about where the search ends, and the single step is taken.
struct alignas(8) PairSlot { uint32_t lt, n_lt; };
struct alignas(8) PairSlot2 { PairSlot a, b; };
constexpr int kPairNone = 16;

// code of stored symbol i (needs the full suffix array); kPairNone for the lone suffix and past the text
static inline MCX_HD int fm_pair_code(const IndexView &ix, uint64_t i, bool &lone)
{
    lone = false;
    if (i >= ix.seq_len) return kPairNone;
    const uint64_t r = i + (i >= ix.primary ? 1 : 0);
    const uint64_t pos = r == 0 ? ix.seq_len : ix.sa_full[r];
    if (pos < 2) { lone = pos == 1; return kPairNone; }
    return ref_code(ix, (int64_t)pos - 1) * 4 + ref_code(ix, (int64_t)pos - 2);
}

// the record of 32 symbols with codes c[0..31]; n_lt[j] (j = 1..15): symbols with a code below j before the record, moved past it
static inline MCX_HD void fm_pair_record(const uint8_t *c, uint64_t n_lt[16], PairSlot out[16])
{
    uint32_t lt = 0;
    for (int j = 1; j < 16; j++) {
        for (int t = 0; t < 32; t++) if (c[t] == j - 1) lt |= 0x80000000u >> t;
        PairSlot e; e.lt = lt; e.n_lt = (uint32_t)n_lt[j];
        if (j <= 8) out[j - 1] = e;
        if (j >= 8) out[j] = e;
        n_lt[j] += (uint64_t)fm_popc(lt);
    }
}

// bwt_occ (bwt_search.cpp:36-51) for k in [0, seq_len]
static inline MCX_HD uint64_t fm_occ(const IndexView &ix, uint64_t k, int c)
{
    if (k == ix.seq_len) return ix.L2[c + 1] - ix.L2[c];
    const uint64_t x = k - (k >= ix.primary);
    FmBlock b;
    fm_load_block(ix.bwt, x >> 7, b);
    return fm_count1(b, (int)(x & 127) + 1, c);
}

// first[s] for s = b1 * 4 + b2: the rows up to and including the last one before the suffixes that begin b2 b1 (the step adds 1 + occ2)
static inline MCX_HD uint64_t fm_pair_first(const IndexView &ix, int s)
{
    const int b1 = s >> 2, b2 = s & 3;
    return ix.L2[b2] + fm_occ(ix, ix.L2[b1], b2);
}

// occ2 and gt2 at stored symbol pos for the code s, modulo 2^32
static inline MCX_HD void fm_pair_counts(const IndexView &ix, uint64_t pos, int s, uint32_t &eq, uint32_t &gt)
{
    const PairSlot *rec = (const PairSlot *)ix.rank2 + (pos >> 5) * 16;
    const int t = s == 0 ? 0 : s < 8 ? s - 1 : s == 15 ? 14 : s;
    const PairSlot2 v = *(const PairSlot2 *)(rec + t);
    const uint32_t m = 0xFFFFFFFFu << (31 - (int)(pos & 31));
    const uint32_t fa = v.a.n_lt + (uint32_t)fm_popc(v.a.lt & m), fb = v.b.n_lt + (uint32_t)fm_popc(v.b.lt & m);
    const uint32_t lone = ix.rank2_lone <= pos ? 1u : 0u;
    const uint32_t all = (uint32_t)(pos + 1) - lone; // below 16: every symbol that has a code
    const uint32_t lo = s == 0 ? 0u : s == 15 ? fb : fa, hi = s == 0 ? fa : s == 15 ? all : fb;
    eq = hi - lo;
    gt = all - hi + ((lone && (s >> 2) <= ix.rank2_t0) ? 1u : 0u);
}

// K-mer jump table (a derived, HBM-resident cache; the search results are unchanged): entry i is
// the bi-interval BWT_Search holds after consuming the K bases spelled by i (first base most
// significant), or x2 = 0 when some extension inside the K-mer comes up empty.  A search that
// starts on a K-mer present in the genome replaces its first K-1 extension steps — the ones with
// wide intervals, two block fetches each — by one 16-byte fetch.  K grows with the text
// (4^K <= seq_len / 2, at most 15), so that the interval a search leaves the table with holds only
// a handful of suffixes and one or two FM steps reach the single-suffix phase.
// Entry: three 40-bit numbers — words 0..2 the low halves of x0, x1, x2, word 3 their bits 32..39.
static inline MCX_HD U4 ktab_pack(uint64_t x0, uint64_t x1, uint64_t x2)
{
    U4 e;
    if (x2 == 0) { e.x = e.y = e.z = e.w = 0; return e; }
    e.x = (uint32_t)x0; e.y = (uint32_t)x1; e.z = (uint32_t)x2;
    e.w = (uint32_t)((x0 >> 32) & 0xFF) | ((uint32_t)((x1 >> 32) & 0xFF) << 8) | ((uint32_t)((x2 >> 32) & 0xFF) << 16);
    return e;
}

static inline MCX_HD bool ktab_lookup(const IndexView &ix, uint32_t idx, uint64_t &x0, uint64_t &x1, uint64_t &x2)
{
    const U4 e = ((const U4 *)ix.ktab)[idx];
    x0 = (uint64_t)e.x | ((uint64_t)(e.w & 0xFF) << 32);
    x1 = (uint64_t)e.y | ((uint64_t)((e.w >> 8) & 0xFF) << 32);
    x2 = (uint64_t)e.z | ((uint64_t)((e.w >> 16) & 0xFF) << 32);
    return x2 != 0;
}

// the K the table is built with for a text of seq_len positions
static inline int ktab_k_for(uint64_t seq_len)
{
    int k = 8;
    while (k < 15 && (4ull << (2 * k)) <= seq_len / 2) k++;
    return k;
}

// what the table holds for one K-mer: BWT_Search's first K-1 extensions (bwt_search.cpp:128-151)
static inline MCX_HD void ktab_entry(const IndexView &ix, uint32_t idx, int K, uint64_t &x0, uint64_t &x1, uint64_t &x2)
{
    int c = (int)((idx >> (2 * (K - 1))) & 3);
    x0 = ix.L2[c] + 1; x1 = ix.L2[3 - c] + 1; x2 = ix.L2[c + 1] - ix.L2[c];
    for (int j = 1; j < K && x2 != 0; j++) {
        c = (int)((idx >> (2 * (K - 1 - j))) & 3);
        uint64_t tk[4], tl[4];
        int nb;
        fm_2occ4(ix, x1 - 1, x1 - 1 + x2, tk, tl, nb);
        const int b = 3 - c;
        const uint64_t n2 = tl[b] - tk[b];
        uint64_t n0 = x0 + ((x1 <= ix.primary && x1 + x2 - 1 >= ix.primary) ? 1 : 0);
        for (int bb = 3; bb > b; bb--) n0 += tl[bb] - tk[bb];
        x0 = n0; x1 = ix.L2[b] + 1 + tk[b]; x2 = n2;
    }
}

// self-check of the pair records, trial t: the bi-interval of a random string of 1..12 bases extended by two random bases — the
// pair step against two single steps in the .bwt blocks (seed_fm's arithmetic on bwt_2occ4's counts)
static inline MCX_HD bool fm_pair_step_agrees(const IndexView &ix, uint64_t t)
{
    uint64_t z = (t + 1) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    const int kk = (int)((z >> 3) % 12) + 1;
    uint64_t x0, x1, x2;
    ktab_entry(ix, (uint32_t)((z >> 8) & ((1ull << (2 * kk)) - 1)), kk, x0, x1, x2);
    if (x2 == 0 || (x2 >> 32)) return true;
    const int b1 = (int)((z >> 40) & 3), b2 = (int)((z >> 42) & 3);
    uint64_t s0 = x0, s1 = x1, s2 = x2;
    bool alive = true;
    for (int step = 0; step < 2 && alive; step++) {
        const int b = step ? b2 : b1;
        uint64_t tk[4], tl[4];
        int nb;
        fm_2occ4(ix, s1 - 1, s1 - 1 + s2, tk, tl, nb);
        const uint64_t n2 = tl[b] - tk[b];
        if (n2 == 0) { alive = false; break; }
        uint64_t n0 = s0 + ((s1 <= ix.primary && s1 + s2 - 1 >= ix.primary) ? 1 : 0);
        for (int bb = 3; bb > b; bb--) n0 += tl[bb] - tk[bb];
        s0 = n0; s1 = ix.L2[b] + 1 + tk[b]; s2 = n2;
    }
    const int s = b1 * 4 + b2;
    uint64_t k = x1 - 1, l = x1 - 1 + x2;
    k -= (k >= ix.primary); l -= (l >= ix.primary);
    uint32_t ek, gk, el, gl;
    fm_pair_counts(ix, k, s, ek, gk);
    fm_pair_counts(ix, l, s, el, gl);
    const uint32_t n2 = el - ek;
    if (!alive) return n2 == 0;
    return n2 == s2 && ix.rank2_c2[s] + 1 + (uint64_t)ek == s1 &&
           x0 + ((x1 <= ix.primary && x1 + x2 - 1 >= ix.primary) ? 1 : 0) + (uint64_t)(uint32_t)(gl - gk) == s0;
}

// set in Hit.len by seed_read for hits whose text position is already known (cleared by the caller
// when it builds the SA task list); not used when the index holds the full suffix array
constexpr int32_t kHitResolved = 1 << 30;

// ---- packed views for the seeding walk ------------------------------------------------------
// The read is packed once into 2-bit words (16 bases per u32, base s of a word at bits 30-2s, the
// layout of the .pac bytes read big-endian) plus an N mask (bit 31-s of word s/32), in a small
// per-lane scratch (LDS on the device, strided so that lanes never share a bank).  After that the
// walk never touches the ASCII read again: the K-mer of a search start, the next base of an
// FM step and 16-base windows for the direct comparison are all shifts of those words.
struct PackedRead {
    uint32_t *w;      // lane's first word; word k at w[k * stride]
    int stride;
    int n_code;       // code words incl. one zero word of slack; the N-mask words follow
};

// words of scratch a read of rlen bases needs
static inline MCX_HD int packed_words(int rlen) { return (rlen + 15) / 16 + 1 + (rlen + 31) / 32 + 1; }

static inline MCX_HD void pack_read(const ReadRef &rd, PackedRead &pk)
{
    ReadCursor cur; cur.chunk = 0; cur.w.x = cur.w.y = cur.w.z = cur.w.w = 0;
    const int rlen = rd.rlen, nc = (rlen + 15) >> 4, nmw = (rlen + 31) >> 5;
    pk.n_code = nc + 1;
    uint32_t *M = pk.w + (size_t)pk.n_code * pk.stride;
    uint32_t cw = 0, nw = 0;
    for (int i = 0; i < rlen; i++) {
        const int c = cursor_code(rd, cur, i);
        cw = (cw << 2) | (uint32_t)(c & 3);
        nw = (nw << 1) | (c > 3 ? 1u : 0u);
        if ((i & 15) == 15) pk.w[(i >> 4) * pk.stride] = cw;
        if ((i & 31) == 31) M[(i >> 5) * pk.stride] = nw;
    }
    if (rlen & 15) pk.w[(rlen >> 4) * pk.stride] = cw << (2 * (16 - (rlen & 15)));
    pk.w[nc * pk.stride] = 0u;
    if (rlen & 31) M[(rlen >> 5) * pk.stride] = (nw << (32 - (rlen & 31))) | (0xFFFFFFFFu >> (rlen & 31)); // past the end counts as N
    M[nmw * pk.stride] = 0xFFFFFFFFu;
}

// 16 bases starting at p <= rlen (MSB first); positions past the read read as 0
static inline MCX_HD uint32_t packed_codes16(const PackedRead &pk, int p)
{
    const int k = p >> 4, sh = (p & 15) * 2;
    const uint32_t hi = pk.w[k * pk.stride], lo = k + 1 < pk.n_code ? pk.w[(k + 1) * pk.stride] : 0u;
    return sh ? (hi << sh) | (lo >> (32 - sh)) : hi;
}

// N flags of the 32 bases starting at p <= rlen (bit 31 = base p); bases past the read end are flagged
static inline MCX_HD uint32_t packed_nmask32(const PackedRead &pk, int p, int rlen)
{
    const uint32_t *M = pk.w + (size_t)pk.n_code * pk.stride;
    const int k = p >> 5, sh = p & 31, last = (rlen + 31) >> 5;
    const uint32_t hi = M[k * pk.stride], lo = k + 1 <= last ? M[(k + 1) * pk.stride] : 0xFFFFFFFFu;
    return sh ? (hi << sh) | (lo >> (32 - sh)) : hi;
}

// symbols f .. f+15 of the forward genome, MSB first (two big-endian words of the .pac bytes)
static inline MCX_HD uint32_t ref_codes16_fwd(const IndexView &ix, int64_t f)
{
    const uint32_t *wp = (const uint32_t *)(ix.pac + ((f >> 4) << 2));
    const int sh = (int)(f & 15) * 2;
    const uint32_t hi = __builtin_bswap32(wp[0]), lo = __builtin_bswap32(wp[1]);
    return sh ? (hi << sh) | (lo >> (32 - sh)) : hi;
}

// 16 symbols of RefSequence starting at 2G-coordinate j (bwt_index.cpp:196-215).  Forward
// strand: a funnel shift of the packed genome.  Reverse strand: T[j+s] = 3 - X[2G-1-j-s], i.e.
// the mirrored forward window with its symbol order reversed, complemented.  Windows that touch
// the strand boundary or the text end are gathered base by base (past the end reads as 0).
static inline MCX_HD uint32_t ref_codes16(const IndexView &ix, int64_t j)
{
    if (j + 16 <= ix.G) return ref_codes16_fwd(ix, j);
    if (j >= ix.G && j + 16 <= ix.G2) {
        uint32_t v = __builtin_bswap32(ref_codes16_fwd(ix, ix.G2 - 16 - j)); // reverse the 16 symbols: bytes,
        v = ((v & 0x0F0F0F0Fu) << 4) | ((v >> 4) & 0x0F0F0F0Fu);             // nibbles,
        v = ((v & 0x33333333u) << 2) | ((v >> 2) & 0x33333333u);             // symbol pairs
        return ~v;
    }
    uint32_t v = 0;
    for (int s = 0; s < 16; s++) v = (v << 2) | (j + s < ix.G2 ? (uint32_t)ref_code(ix, j + s) : 0u);
    return v;
}

// Greedy left-to-right seeding of one read: IdentifySimplePairs (ReadMapping.cpp:125-158)
// driving BWT_Search (bwt_search.cpp:121-164).  Hits are written as BWT rows (x0 + i), to be
// resolved by the SA kernel, or directly as text positions (kHitResolved).  Returns the number of
// hits the read produced (may exceed cap: overflow, nothing lost yet because the pair is then
// re-run in the next tier).
//
// Three phases per search, each as cheap as it can be made without changing a result:
//  1. start: if the next ktab_k bases hold no N and that k-mer occurs in the text, its bi-interval
//     comes from the jump table (one 16-byte fetch instead of ktab_k - 1 wide-interval steps);
//  2. FM steps (one or two 64-byte block fetches each) while the interval holds several suffixes;
//  3. once it holds exactly one (x2 == 1) the pattern has one occurrence in the text, so "can it
//     be extended by base c" is "is the next text base c": the suffix is resolved to its text
//     position (one suffix-array fetch) and the rest of the search is a comparison of 16-base
//     windows of the packed read against the 2-bit genome.  Same length, same position.
// (prepacked: pk already holds the read's words — the device packs whole batches in one pass)
// The walk comes in two pieces so that the device can hand a lane its next read as soon as it has
// finished one (k_seed): seed_next_start, seed_search; seed_read is their plain loop.
//
// next search start at or after p: skips N (ReadMapping.cpp:135).  False when the read has no further search;
// nm: N flags of the 32 bases from the start on.
static inline MCX_HD bool seed_next_start(const PackedRead &pk, int rlen, int &p, uint32_t &nm)
{
    const int stop = rlen - kMinSeedLength;
    nm = 0;
    while (p < stop) {
        nm = packed_nmask32(pk, p, rlen);
        if (!(nm & 0x80000000u)) break;
        p += nm == 0xFFFFFFFFu ? 32 : __builtin_clz(~nm);
    }
    return p < stop;
}

// One search of the greedy walk, from start p (found by seed_next_start), in resumable pieces: its three phases as loops of
// their own, each with a budget of steps.  The lanes of a wave sit at unrelated points of their reads; with this shape they
// run the same phase at the same time (jump-table fetches together, FM steps together, window comparisons together) instead
// of serialising each other's phases, and a lane deep inside a repeat (dozens of FM steps with a wide interval) holds the
// others up for one budget at a time, not for its whole walk: they go on to their comparison phase, their hits and their
// next search while it continues where it stopped.
struct SeedWalk {
    uint64_t x0, x1, x2;   // the bi-interval (BWT_Search's ik)
    int64_t tpos;          // phase 2: text position of the one suffix left
    uint32_t carry;        // phase 2: the packed genome word two consecutive windows share
    int32_t carry_dir;     // 0: none, +1: forward strand (carry = the next window's high word), -1: reverse strand (its low word)
    int32_t start;         // read position the search began at
    int32_t phase;         // 0: no search under way, 1: FM steps, 2: window comparison, 3: over (hits to be taken)
    int32_t ended;         // the FM phase met an N, the read's end or an empty extension: no comparison phase
};

// phase 0 -> 1: the start of a search at p — the jump table's entry when the next ktab_k bases hold no N and occur in the text
static inline MCX_HD void seed_begin(const IndexView &ix, const PackedRead &pk, int rlen, uint32_t nm, int &p, SeedWalk &w)
{
    w.start = p; w.ended = 0; w.tpos = 0; w.carry = 0; w.carry_dir = 0;
    const uint32_t c16 = packed_codes16(pk, p);
    bool jumped = false;
    if (ix.ktab && p + ix.ktab_k <= rlen && (nm >> (32 - ix.ktab_k)) == 0)
        if (ktab_lookup(ix, c16 >> (32 - 2 * ix.ktab_k), w.x0, w.x1, w.x2)) { p += ix.ktab_k; jumped = true; }
    if (!jumped) { const int c = (int)(c16 >> 30); w.x0 = ix.L2[c] + 1; w.x1 = ix.L2[3 - c] + 1; w.x2 = ix.L2[c + 1] - ix.L2[c]; p++; }
    w.phase = 1;
}

// phase 1: FM steps while the interval holds several suffixes (or none: the step then ends the search); at most max_steps of them
static inline MCX_HD void seed_fm(const IndexView &ix, const PackedRead &pk, int rlen, int &p, SeedWalk &w, int64_t &blocks, int max_steps)
{
    int steps = 0;
    while (w.x2 != 1 && !w.ended && steps < max_steps) {
        steps++;
        const uint32_t nm2 = packed_nmask32(pk, p, rlen);
        if (nm2 & 0x80000000u) { w.ended = 1; break; } // N or read end
        const uint32_t c16 = packed_codes16(pk, p);
        const int c = (int)(c16 >> 30);
        const int b = 3 - c;
        bool last = false; // the step after this one is known to come up empty
        if (ix.rank2 && !(nm2 & 0x40000000u) && (w.x2 >> 32) == 0) { // this base and the next in one step (fm_pair_counts)
            const int s = b * 4 + 3 - (int)((c16 >> 28) & 3);
            uint64_t k = w.x1 - 1, l = w.x1 - 1 + w.x2;
            k -= (k >= ix.primary); l -= (l >= ix.primary);
            uint32_t ek, gk, el, gl;
            fm_pair_counts(ix, k, s, ek, gk);
            fm_pair_counts(ix, l, s, el, gl);
            blocks += (k >> 5) != (l >> 5) ? 2 : 1;
            const uint32_t n2 = el - ek;
            if (n2 != 0) {
                w.x0 = w.x0 + ((w.x1 <= ix.primary && w.x1 + w.x2 - 1 >= ix.primary) ? 1 : 0) + (uint64_t)(uint32_t)(gl - gk);
                w.x1 = ix.rank2_c2[s] + 1 + (uint64_t)ek; w.x2 = n2;
                p += 2;
                continue;
            }
            last = true;
        }
        uint64_t tkb, n2, above; // occ(k, b); occ(l, b) - occ(k, b); the same difference summed over the bases above b
        if (ix.rank) {
            uint64_t k = w.x1 - 1, l = w.x1 - 1 + w.x2;
            k -= (k >= ix.primary); l -= (l >= ix.primary);
            const RankChunk *R = (const RankChunk *)ix.rank + (uint64_t)b * ix.rank_chunks;
            const uint64_t ck = k >> 5, cl = l >> 5;
            const RankChunk rk = R[ck];
            RankChunk rl = rk;
            if (cl != ck) rl = R[cl];
            blocks += cl != ck ? 2 : 1;
            const uint64_t x0 = ix.rank_cross[0], x1 = ix.rank_cross[1], x2 = ix.rank_cross[2], x3 = ix.rank_cross[3];
            const uint64_t y0 = ix.rank_cross[4], y1 = ix.rank_cross[5], y2 = ix.rank_cross[6], y3 = ix.rank_cross[7];
            const uint64_t xe = b == 0 ? x0 : b == 1 ? x1 : b == 2 ? x2 : x3, xg = b == 0 ? y0 : b == 1 ? y1 : b == 2 ? y2 : y3;
            const uint32_t mk = 0xFFFFFFFFu << (31 - (int)(k & 31)), ml = 0xFFFFFFFFu << (31 - (int)(l & 31));
            const uint64_t ek = (uint64_t)rk.n_eq + (uint32_t)fm_popc(rk.eq & mk) + (ck >= xe ? 1ull << 32 : 0ull);
            const uint64_t el = (uint64_t)rl.n_eq + (uint32_t)fm_popc(rl.eq & ml) + (cl >= xe ? 1ull << 32 : 0ull);
            const uint64_t gk = (uint64_t)rk.n_gt + (uint32_t)fm_popc(rk.gt & mk) + (ck >= xg ? 1ull << 32 : 0ull);
            const uint64_t gl = (uint64_t)rl.n_gt + (uint32_t)fm_popc(rl.gt & ml) + (cl >= xg ? 1ull << 32 : 0ull);
            tkb = ek; n2 = el - ek; above = gl - gk;
        } else {
            uint64_t tk[4], tl[4];
            int nb;
            fm_2occ4(ix, w.x1 - 1, w.x1 - 1 + w.x2, tk, tl, nb);
            blocks += nb;
            tkb = tk[b]; n2 = tl[b] - tk[b]; above = 0;
            for (int bb = 3; bb > b; bb--) above += tl[bb] - tk[bb];
        }
        if (n2 == 0) { w.ended = 1; break; }
        // ok[3].x0 = ik.x0 + primary correction; lower bases stack on top (:143-146)
        const uint64_t n0 = w.x0 + ((w.x1 <= ix.primary && w.x1 + w.x2 - 1 >= ix.primary) ? 1 : 0) + above;
        w.x0 = n0; w.x1 = ix.L2[b] + 1 + tkb; w.x2 = n2;
        p++;
        if (last) { w.ended = 1; break; }
    }
    if (w.ended) w.phase = 3;
    else if (w.x2 == 1) { // exactly one suffix left: the rest of the search is a comparison with the text itself
        int lf = 0;
        w.tpos = (int64_t)fm_sa(ix, w.x0, lf);
        w.carry = 0; w.carry_dir = 0;
        w.phase = 2;
    }
}

// phase 2: the pattern has one occurrence in the text, so "can it be extended by base c" is "is the next text base c":
// 16-base windows of the packed read against the 2-bit genome, at most max_windows of them.  Consecutive windows overlap by
// one packed genome word: while the walk advances by whole windows the word is carried over, so each further window costs ONE
// 4-byte fetch.
static inline MCX_HD void seed_compare(const IndexView &ix, const PackedRead &pk, int rlen, int &p, SeedWalk &w, int max_windows)
{
    for (int k = 0; k < max_windows; k++) {
        const int64_t j = w.tpos + (p - w.start);
        int64_t room = (int64_t)ix.seq_len - j;
        if (rlen - p < room) room = rlen - p;
        if (room <= 0) { w.phase = 3; return; }
        uint32_t ref;
        if (j + 16 <= ix.G) { // forward strand: funnel shift of two big-endian words of the .pac bytes
            const uint32_t *wp = (const uint32_t *)ix.pac + (j >> 4);
            const int sh = (int)(j & 15) * 2;
            const uint32_t hi = w.carry_dir > 0 ? w.carry : __builtin_bswap32(wp[0]), lo = __builtin_bswap32(wp[1]);
            ref = sh ? (hi << sh) | (lo >> (32 - sh)) : hi;
            w.carry = lo; w.carry_dir = 1;
        } else if (j >= ix.G && j + 16 <= ix.G2) { // reverse strand: the mirrored forward window, reversed and complemented
            const int64_t f = ix.G2 - 16 - j;
            const uint32_t *wp = (const uint32_t *)ix.pac + (f >> 4);
            const int sh = (int)(f & 15) * 2;
            const uint32_t hi = __builtin_bswap32(wp[0]), lo = w.carry_dir < 0 ? w.carry : __builtin_bswap32(wp[1]);
            uint32_t v = __builtin_bswap32(sh ? (hi << sh) | (lo >> (32 - sh)) : hi);
            v = ((v & 0x0F0F0F0Fu) << 4) | ((v >> 4) & 0x0F0F0F0Fu);
            v = ((v & 0x33333333u) << 2) | ((v >> 2) & 0x33333333u);
            ref = ~v;
            w.carry = hi; w.carry_dir = -1;
        } else { ref = ref_codes16(ix, j); w.carry_dir = 0; }
        const uint32_t x = packed_codes16(pk, p) ^ ref;
        uint32_t sp = packed_nmask32(pk, p, rlen) >> 16; // N flags, bit 15-s -> bit 30-2s
        sp = (sp | (sp << 8)) & 0x00FF00FFu; sp = (sp | (sp << 4)) & 0x0F0F0F0Fu;
        sp = (sp | (sp << 2)) & 0x33333333u; sp = (sp | (sp << 1)) & 0x55555555u;
        const uint32_t mm = ((x | (x >> 1)) & 0x55555555u) | sp; // bit 30-2s: base s differs or is N
        int same = mm ? (__builtin_clz(mm) >> 1) : 16;
        if (same > room) same = (int)room;
        p += same;
        if (same < 16) { w.phase = 3; return; }
    }
}

// The same phase with 64 bases of the genome per fetch instead of 16: two aligned 16-byte chunks of the 2-bit genome (128 bases)
// hold the next 64 text positions wherever they begin, so a seed of 150 bases is confirmed in three dependent fetches, not ten —
// and it is dependent fetches that the seeding kernel's time is made of.  Stretches that touch the strand boundary or the end
// of the text take seed_compare's single windows.  Same comparisons, same p.
static inline MCX_HD void seed_compare_wide64(const IndexView &ix, const PackedRead &pk, int rlen, int &p, SeedWalk &w, int max_fetches)
{
    for (int k = 0; k < max_fetches; k++) {
        const int64_t j = w.tpos + (p - w.start);
        const bool fwd = j + 64 <= ix.G, rev = j >= ix.G && j + 64 <= ix.G2;
        if (!fwd && !rev) {
            w.carry_dir = 0;
            seed_compare(ix, pk, rlen, p, w, 1);
            if (w.phase == 3) return;
            continue;
        }
        const int64_t f_lo = fwd ? j : ix.G2 - 64 - j;   // the forward stretch [f_lo, f_lo + 64) under the 64 text positions
        const U4 *src = (const U4 *)ix.pac + (f_lo >> 6);
        const U4 a = src[0], b = src[1];
        const int oq = (int)(f_lo & 63) >> 4, sh = (int)(f_lo & 15) * 2;
        const uint32_t W0 = __builtin_bswap32(a.x), W1 = __builtin_bswap32(a.y), W2 = __builtin_bswap32(a.z), W3 = __builtin_bswap32(a.w);
        const uint32_t W4 = __builtin_bswap32(b.x), W5 = __builtin_bswap32(b.y), W6 = __builtin_bswap32(b.z), W7 = __builtin_bswap32(b.w);
        uint32_t V[5]; // the five words the stretch lies in
        V[0] = oq == 0 ? W0 : oq == 1 ? W1 : oq == 2 ? W2 : W3;
        V[1] = oq == 0 ? W1 : oq == 1 ? W2 : oq == 2 ? W3 : W4;
        V[2] = oq == 0 ? W2 : oq == 1 ? W3 : oq == 2 ? W4 : W5;
        V[3] = oq == 0 ? W3 : oq == 1 ? W4 : oq == 2 ? W5 : W6;
        V[4] = oq == 0 ? W4 : oq == 1 ? W5 : oq == 2 ? W6 : W7;
        MCX_UNROLL
        for (int t = 0; t < 4; t++) {
            const int64_t jj = w.tpos + (p - w.start); // (= j + 16 t: every window before this one was whole)
            int64_t room = (int64_t)ix.seq_len - jj;
            if (rlen - p < room) room = rlen - p;
            if (room <= 0) { w.phase = 3; return; }
            const uint32_t vh = fwd ? V[t] : V[3 - t], vl = fwd ? V[t + 1] : V[4 - t]; // text window t is forward window 3 - t on the reverse strand, mirrored
            uint32_t ref = sh ? (vh << sh) | (vl >> (32 - sh)) : vh;
            if (!fwd) {
                uint32_t v = __builtin_bswap32(ref);
                v = ((v & 0x0F0F0F0Fu) << 4) | ((v >> 4) & 0x0F0F0F0Fu);
                v = ((v & 0x33333333u) << 2) | ((v >> 2) & 0x33333333u);
                ref = ~v;
            }
            const uint32_t x = packed_codes16(pk, p) ^ ref;
            uint32_t sp = packed_nmask32(pk, p, rlen) >> 16; // N flags, bit 15-s -> bit 30-2s
            sp = (sp | (sp << 8)) & 0x00FF00FFu; sp = (sp | (sp << 4)) & 0x0F0F0F0Fu;
            sp = (sp | (sp << 2)) & 0x33333333u; sp = (sp | (sp << 1)) & 0x55555555u;
            const uint32_t mm = ((x | (x >> 1)) & 0x55555555u) | sp;
            int same = mm ? (__builtin_clz(mm) >> 1) : 16;
            if (same > room) same = (int)room;
            p += same;
            if (same < 16) { w.phase = 3; return; }
        }
    }
}

// The same again, with the chunk two consecutive fetches share kept in registers: the first fetch of a search brings 64 bases (two
// chunks wherever the stretch begins), every further one 128 — the chunk kept from the last fetch and the two beyond it (before it,
// on the reverse strand, where the text runs against the genome) — so a seed of 150 bases is confirmed in two dependent fetches,
// and a fetch asks for the line its predecessor already brought half as often.  Same comparisons, same p.
static inline MCX_HD void seed_compare_wide(const IndexView &ix, const PackedRead &pk, int rlen, int &p, SeedWalk &w, int max_fetches)
{
    bool have = false;   // `keep` holds the chunk the next stretch begins in (forward) / ends in (reverse)
    U4 keep; keep.x = keep.y = keep.z = keep.w = 0u;
    int64_t keep_at = 0; // its index among the genome's 16-byte chunks
    for (int k = 0; k < max_fetches; k++) {
        const int64_t j = w.tpos + (p - w.start);
        const int nb = have ? 128 : 64;
        const bool fwd = j + nb <= ix.G, rev = j >= ix.G && j + nb <= ix.G2;
        if (!fwd && !rev) {
            if (have) { have = false; continue; } // (128 bases do not fit any more: 64 may)
            w.carry_dir = 0;
            seed_compare(ix, pk, rlen, p, w, 1);
            if (w.phase == 3) return;
            continue;
        }
        const int64_t f_lo = fwd ? j : ix.G2 - nb - j;   // the forward stretch [f_lo, f_lo + nb) under the nb text positions
        const int64_t at = f_lo >> 6;
        if (have && keep_at != (fwd ? at : at + 2)) { have = false; continue; }
        const U4 *src = (const U4 *)ix.pac + at;
        // twelve words X0..X11 of three consecutive chunks; forward window u of the stretch lies in X[base + oq + u], X[base + oq + u + 1]:
        //   forward, kept chunk:  X0-3 = keep, X4-11 = the two chunks behind it           reverse, kept chunk:  X0-7 = the two chunks before it, X8-11 = keep
        //   forward, first fetch: X0-7 = the two chunks, X8-11 unused                     reverse, first fetch: X4-11 = the two chunks (so that text
        //                                                                                 window t is V[7 - t], V[8 - t] either way), X0-3 unused
        const U4 l0 = src[fwd && have ? 1 : 0], l1 = src[fwd && have ? 2 : 1];
        U4 c0, c1, c2;
        if (fwd) { if (have) { c0 = keep; c1 = l0; c2 = l1; } else { c0 = l0; c1 = l1; c2 = l1; } }
        else { if (have) { c0 = l0; c1 = l1; c2 = keep; } else { c0 = l0; c1 = l0; c2 = l1; } }
        const uint32_t X0 = __builtin_bswap32(c0.x), X1 = __builtin_bswap32(c0.y), X2 = __builtin_bswap32(c0.z), X3 = __builtin_bswap32(c0.w);
        const uint32_t X4 = __builtin_bswap32(c1.x), X5 = __builtin_bswap32(c1.y), X6 = __builtin_bswap32(c1.z), X7 = __builtin_bswap32(c1.w);
        const uint32_t X8 = __builtin_bswap32(c2.x), X9 = __builtin_bswap32(c2.y), X10 = __builtin_bswap32(c2.z), X11 = __builtin_bswap32(c2.w);
        const int oq = (int)(f_lo & 63) >> 4, sh = (int)(f_lo & 15) * 2;
        uint32_t V[9]; // V[u] = X[oq + u]
        V[0] = oq == 0 ? X0 : oq == 1 ? X1 : oq == 2 ? X2 : X3;
        V[1] = oq == 0 ? X1 : oq == 1 ? X2 : oq == 2 ? X3 : X4;
        V[2] = oq == 0 ? X2 : oq == 1 ? X3 : oq == 2 ? X4 : X5;
        V[3] = oq == 0 ? X3 : oq == 1 ? X4 : oq == 2 ? X5 : X6;
        V[4] = oq == 0 ? X4 : oq == 1 ? X5 : oq == 2 ? X6 : X7;
        V[5] = oq == 0 ? X5 : oq == 1 ? X6 : oq == 2 ? X7 : X8;
        V[6] = oq == 0 ? X6 : oq == 1 ? X7 : oq == 2 ? X8 : X9;
        V[7] = oq == 0 ? X7 : oq == 1 ? X8 : oq == 2 ? X9 : X10;
        V[8] = oq == 0 ? X8 : oq == 1 ? X9 : oq == 2 ? X10 : X11;
        const int n_win = nb >> 4;
        MCX_UNROLL
        for (int t = 0; t < 8; t++) {
            if (t >= n_win) break;
            const int64_t jj = w.tpos + (p - w.start); // (= j + 16 t: every window before this one was whole)
            int64_t room = (int64_t)ix.seq_len - jj;
            if (rlen - p < room) room = rlen - p;
            if (room <= 0) { w.phase = 3; return; }
            const uint32_t vh = fwd ? V[t] : V[7 - t], vl = fwd ? V[t + 1] : V[8 - t]; // on the reverse strand text window t is the stretch's last forward window but t, mirrored
            uint32_t ref = sh ? (vh << sh) | (vl >> (32 - sh)) : vh;
            if (!fwd) {
                uint32_t v = __builtin_bswap32(ref);
                v = ((v & 0x0F0F0F0Fu) << 4) | ((v >> 4) & 0x0F0F0F0Fu);
                v = ((v & 0x33333333u) << 2) | ((v >> 2) & 0x33333333u);
                ref = ~v;
            }
            const uint32_t x = packed_codes16(pk, p) ^ ref;
            uint32_t sp = packed_nmask32(pk, p, rlen) >> 16; // N flags, bit 15-s -> bit 30-2s
            sp = (sp | (sp << 8)) & 0x00FF00FFu; sp = (sp | (sp << 4)) & 0x0F0F0F0Fu;
            sp = (sp | (sp << 2)) & 0x33333333u; sp = (sp | (sp << 1)) & 0x55555555u;
            const uint32_t mm = ((x | (x >> 1)) & 0x55555555u) | sp;
            int same = mm ? (__builtin_clz(mm) >> 1) : 16;
            if (same > room) same = (int)room;
            p += same;
            if (same < 16) { w.phase = 3; return; }
        }
        // every window was whole: the next stretch begins nb positions on — in the last chunk fetched (forward), ends in the first (reverse)
        keep = fwd ? l1 : l0; keep_at = fwd ? at + (have ? 2 : 1) : at; have = true;
    }
}

// phase 3 -> 0: the search is over — its hits (BWT_Search's len >= MinSeedLength && freq <= OCC_Thr), p at the next candidate start
static inline MCX_HD void seed_take(const IndexView &ix, int &p, SeedWalk &w, Hit *hits, int cap, int &n_hits, int64_t &ext_steps)
{
    const int len = p - w.start;
    ext_steps += len;
    if (len >= kMinSeedLength && w.x2 <= (uint64_t)kOccThr) {
        // with every suffix-array entry in memory a row resolves with one fetch, here and now: all hits
        // then leave as text positions, unflagged, and nothing is left for the SA pass
        const bool direct = ix.sa_full != nullptr;
        if (w.x2 == 1 && !w.ended) {
            if (n_hits < cap) { Hit h; h.gPos = w.tpos; h.rPos = w.start; h.len = direct ? len : (len | kHitResolved); hits[n_hits] = h; }
            n_hits++;
        } else if (direct) {
            // the rows of the interval lie side by side in the suffix array: fetched as 16-byte pairs, four pairs at a time, every
            // fetch of a round issued before the first hit is stored (row by row each fetch would wait for the store before it:
            // a chain of up to fifty round trips in one lane, with the rest of the wave looking on)
            const U4 *pairs = (const U4 *)ix.sa_full;
            const uint64_t end = w.x0 + w.x2;
            for (uint64_t c = w.x0 >> 1; 2 * c < end; c += 4) {
                U4 v0 = pairs[c], v1 = v0, v2 = v0, v3 = v0;
                if (2 * (c + 1) < end) v1 = pairs[c + 1];
                if (2 * (c + 2) < end) v2 = pairs[c + 2];
                if (2 * (c + 3) < end) v3 = pairs[c + 3];
                const uint64_t row[8] = {(uint64_t)v0.x | ((uint64_t)v0.y << 32), (uint64_t)v0.z | ((uint64_t)v0.w << 32), (uint64_t)v1.x | ((uint64_t)v1.y << 32), (uint64_t)v1.z | ((uint64_t)v1.w << 32),
                                         (uint64_t)v2.x | ((uint64_t)v2.y << 32), (uint64_t)v2.z | ((uint64_t)v2.w << 32), (uint64_t)v3.x | ((uint64_t)v3.y << 32), (uint64_t)v3.z | ((uint64_t)v3.w << 32)};
                MCX_UNROLL
                for (int t = 0; t < 8; t++) {
                    const uint64_t r = 2 * c + (uint64_t)t;
                    if (r < w.x0 || r >= end) continue;
                    if (n_hits < cap) { Hit h; h.gPos = (int64_t)row[t]; h.rPos = w.start; h.len = len; hits[n_hits] = h; }
                    n_hits++;
                }
            }
        } else for (uint64_t i = 0; i < w.x2; i++) {
            if (n_hits < cap) { Hit h; h.gPos = (int64_t)(w.x0 + i); h.rPos = w.start; h.len = len; hits[n_hits] = h; }
            n_hits++;
        }
    }
    p = p + 1;
    w.phase = 0;
}

// one whole search (the host emulation's form)
static inline MCX_HD void seed_search(const IndexView &ix, const PackedRead &pk, int rlen, uint32_t nm, int &p, Hit *hits, int cap,
                                      int &n_hits, int64_t &ext_steps, int64_t &blocks)
{
    SeedWalk w;
    seed_begin(ix, pk, rlen, nm, p, w);
    while (w.phase == 1) seed_fm(ix, pk, rlen, p, w, blocks, 1 << 30);
    while (w.phase == 2) seed_compare_wide(ix, pk, rlen, p, w, 1 << 30);
    seed_take(ix, p, w, hits, cap, n_hits, ext_steps);
}

static inline MCX_HD int seed_read(const IndexView &ix, const ReadRef &rd, PackedRead pk, Hit *hits, int cap,
                                   int64_t &ext_steps, int64_t &blocks, bool prepacked = false)
{
    const int rlen = rd.rlen;
    if (!prepacked) pack_read(rd, pk);
    int n_hits = 0, p = 0;
    uint32_t nm;
    while (seed_next_start(pk, rlen, p, nm)) seed_search(ix, pk, rlen, nm, p, hits, cap, n_hits, ext_steps, blocks);
    return n_hits;
}

} // namespace mcx
#endif
