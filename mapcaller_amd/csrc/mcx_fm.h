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

// keep-mask of word j for a count over the first n (1..128) symbols: words before the one that
// holds symbol n-1 are kept whole, that word keeps its top `rem` symbols, later words nothing
static inline MCX_HD uint32_t fm_word_mask(int j, int q, uint32_t last_mask)
{
    return j < q ? 0x55555555u : (j == q ? last_mask : 0u);
}

// occurrences of each base among the first n (1..128) symbols of the block, added to occ[]
static inline MCX_HD void fm_count4(const FmBlock &b, int n, uint64_t cnt[4])
{
    const int q = (n - 1) >> 4, rem = ((n - 1) & 15) + 1;
    const uint32_t last_mask = 0x55555555u & (0xFFFFFFFFu << (32 - 2 * rem));
    uint32_t t = 0, ct = 0, gt = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t mask = fm_word_mask(j, q, last_mask);
        const uint32_t lo = b.w[j] & mask, hi = (b.w[j] >> 1) & mask;
        t += (uint32_t)__builtin_popcount(hi & lo);
        ct += (uint32_t)__builtin_popcount(lo);
        gt += (uint32_t)__builtin_popcount(hi);
    }
    cnt[0] = b.occ[0] + ((uint32_t)n - ct - gt + t);
    cnt[1] = b.occ[1] + (ct - t);
    cnt[2] = b.occ[2] + (gt - t);
    cnt[3] = b.occ[3] + t;
}

// occurrences of base c among the first n symbols (bwt_occ, bwt_search.cpp:25-47)
static inline MCX_HD uint64_t fm_count1(const FmBlock &b, int n, int c)
{
    const int q = (n - 1) >> 4, rem = ((n - 1) & 15) + 1;
    const uint32_t last_mask = 0x55555555u & (0xFFFFFFFFu << (32 - 2 * rem));
    uint32_t s = 0;
    const uint32_t xlo = (c & 1) ? 0u : 0xFFFFFFFFu, xhi = (c & 2) ? 0u : 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t mask = fm_word_mask(j, q, last_mask);
        const uint32_t lo = b.w[j] ^ xlo, hi = (b.w[j] >> 1) ^ xhi;
        s += (uint32_t)__builtin_popcount(lo & hi & mask);
    }
    return b.occ[c] + s;
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
        const int c = (int)(packed_codes16(pk, p) >> 30);
        uint64_t tk[4], tl[4];
        int nb;
        fm_2occ4(ix, w.x1 - 1, w.x1 - 1 + w.x2, tk, tl, nb);
        blocks += nb;
        const int b = 3 - c;
        const uint64_t n2 = tl[b] - tk[b];
        if (n2 == 0) { w.ended = 1; break; }
        // ok[3].x0 = ik.x0 + primary correction; lower bases stack on top (:143-146)
        uint64_t n0 = w.x0 + ((w.x1 <= ix.primary && w.x1 + w.x2 - 1 >= ix.primary) ? 1 : 0);
        for (int bb = 3; bb > b; bb--) n0 += tl[bb] - tk[bb];
        w.x0 = n0; w.x1 = ix.L2[b] + 1 + tk[b]; w.x2 = n2;
        p++;
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
        } else for (uint64_t i = 0; i < w.x2; i++) {
            if (n_hits < cap) { Hit h; h.gPos = direct ? (int64_t)ix.sa_full[w.x0 + i] : (int64_t)(w.x0 + i); h.rPos = w.start; h.len = len; hits[n_hits] = h; }
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
    while (w.phase == 2) seed_compare(ix, pk, rlen, p, w, 1 << 30);
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
