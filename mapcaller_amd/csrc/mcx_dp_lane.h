// mapcaller_amd/csrc/mcx_dp_lane.h — gapped extension, ONE PROBLEM PER LANE.
//
// Same recurrences and the same traceback as mcx_dp.h (ksw_extz2_sse + ksw_backtrack, reference
// src/ksw2_alignment.cpp:25-248; nw_alignment, src/nw_alignment.cpp:18-83), organised for what the path
// actually produces: millions of small problems per batch (BASELINE config 5: 26 M per 8 M reads,
// mean 19 x 17 cells).  A wavefront that sweeps the anti-diagonals of ONE such problem leaves most of
// its lanes idle and spends its instructions on moving neighbours' values between lanes; here every
// lane owns a problem, so 64 problems advance per instruction and nothing crosses lanes:
//
//  * the matrix is swept in vertical STRIPS of K target columns, row by row: the values a cell needs
//    from the row above (nw: S, T; ksw2: u, y) live in 2K registers, the values from its left
//    neighbour (nw: R, S; ksw2: x, v) are carried along the row in two more; the right edge of a
//    strip is parked per row in one word for the next strip (only problems wider than K use it);
//  * the anti-diagonal order of the reference is a vectorisation device, not part of the result:
//    every cell depends on its left, upper and upper-left neighbours only, so the row-major sweep
//    computes the same cells — and the same traceback flags — exactly;
//  * traceback flags are packed (nw: 2 bits per cell, ksw2: 4) — a row of a strip is one or two
//    words, 8-16 x smaller than a byte per cell of the (q + t - 1) x t band — and stored lane-
//    interleaved: word w of lane l at [w * 64 + l], so the wave's stores and loads are whole lines;
//  * the lane then walks its own traceback (the reference's walk, unchanged) and leaves the column
//    string and its DpSummary.
//
// LaneMem abstracts "word w of this lane" so that the same code runs on the host (stride 1) in the
// CPU-side tests, against the reference's own vectors.
#ifndef MCX_DP_LANE_H
#define MCX_DP_LANE_H
#include "mcx_glue.h"

namespace mcx {

// The traceback lane fills the problem's DpSummary on its way from the last column to the first (out null: no summary).
struct DpSumAcc {
    DpSummary *out;
    int n, mis, switches, cur, run, n_rle, pd, pi, pr, td, ti, tr;
    bool seen_m;
    MCX_HD void begin(DpSummary *o) { out = o; n = mis = switches = run = n_rle = pd = pi = pr = td = ti = tr = 0; cur = -1; seen_m = false; }
    MCX_HD void flush()
    {
        if (run > 0) { if (n_rle < kDpRle) out->rle[kDpRle - 1 - n_rle] = ((uint32_t)run << 4) | (uint32_t)cur; n_rle++; }
    }
    // k: 0 'M', 1 'I', 2 'D' (the CIGAR codes); differ: an 'M' column over two different bases
    MCX_HD void put(int k, int differ)
    {
        if (!out) return;
        if (k != cur) { flush(); cur = k; run = 0; switches++; if (k) pr++; }
        run++;
        if (k == 0) {
            n++; mis += differ;
            if (!seen_m) { td = pd; ti = pi; tr = pr; seen_m = true; } // what came before the walk's first 'M' is the string's tail
            pd = pi = pr = 0;
        } else if (k == 2) pd++; else pi++;
    }
    MCX_HD void end(uint32_t cols_off, int cols_len)
    {
        if (!out) return;
        flush();
        if (!seen_m) { td = pd; ti = pi; tr = pr; }
        out->cols_off = cols_off; out->cols_len = (uint16_t)cols_len;
        out->n = (uint16_t)n; out->mis = (uint16_t)mis; out->switches = (uint16_t)switches;
        out->lead_d = (uint16_t)pd; out->lead_i = (uint16_t)pi; out->lead_runs = (uint16_t)pr; // what is pending at the string's start is its head
        out->tail_d = (uint16_t)td; out->tail_i = (uint16_t)ti; out->tail_runs = (uint16_t)tr;
        out->n_rle = n_rle <= kDpRle ? (uint16_t)n_rle : (uint16_t)0xFFFF;
    }
};

struct LaneMem {
    uint32_t *base;  // word 0 of lane 0: the same for every lane of the wavefront, so that a fetch is "scalar base + 32-bit byte offset" — one register and one
                     // instruction per address where a per-lane pointer took two of each (a window fetch of a walk holds a dozen addresses at once)
    uint32_t stride; // words between consecutive words of a lane (64 on the device: lanes interleaved; 1 on the host)
    uint32_t lane;   // this lane's place among them (0 on the host)
    MCX_HD uint32_t get(uint32_t w) const { return *(const uint32_t *)((const char *)base + (w * stride + lane) * 4u); } // (a stretch is a few MB: the offset fits)
    MCX_HD void put(uint32_t w, uint32_t v) const { *(uint32_t *)((char *)base + (w * stride + lane) * 4u) = v; }
};

// where a group of problems keeps its words (the same for every lane of the wave: offsets come from the group's largest problem)
struct LaneLayout {
    uint32_t off_q;    // query: per 16 rows one word of 2-bit codes (row 0 in the top bits) and one of N flags (bit 15 = row 0)
    uint32_t off_edge; // one word per row: the strip's right edge (nw: R | S << 16; ksw2: x | (v + 8) << 8)
    uint32_t off_dir;  // traceback words: ((strip * rows + row) * DW + k)
    uint32_t rows;     // row pitch of the traceback (the group's longest query)
    uint32_t words;    // total
};

template <int K, bool NW> struct LaneDir { static constexpr int bits = NW ? 2 : 4, words = (K * bits + 31) / 32; };

template <int K, bool NW>
static inline MCX_HD LaneLayout lane_layout(int rows, int strips)
{
    LaneLayout l;
    l.rows = (uint32_t)rows;
    l.off_q = 0;
    l.off_edge = 2u * (uint32_t)((rows + 15) >> 4);
    l.off_dir = l.off_edge + (uint32_t)rows;
    l.words = l.off_dir + (uint32_t)strips * (uint32_t)rows * (uint32_t)LaneDir<K, NW>::words;
    return l;
}

// The query of a problem into the lane's words.  get16(p): 16 codes (2 bits each, first in the top bits) + 16 N flags (bit 15 first)
// of the problem's query positions p .. p+15; positions past the end may hold anything.
template <class Get16>
static inline MCX_HD void lane_stage_query(const LaneMem &m, const LaneLayout &l, int qlen, Get16 get16)
{
    for (int p = 0, w = 0; p < qlen; p += 16, w++) {
        uint32_t codes, flags;
        get16(p, codes, flags);
        m.put(l.off_q + 2 * w, codes);
        m.put(l.off_q + 2 * w + 1, flags);
    }
}

// ---------------------------------------------------------------------------------------------
// nw (doubled integers, equality-based traceback flags: bit 0 = "s == r" (a 'D' column), bit 1 = "s == t" ('I'))
// tgt16(b0): the target's codes b0 .. b0+15 as 2-bit fields, first in the top bits (the genome holds no N).
// Returns the final score s[m][n] (doubled, as dp_nw_core).
// ---------------------------------------------------------------------------------------------
template <int K, class Tgt16>
static inline MCX_HD int lane_sweep_nw(const LaneMem &mem, const LaneLayout &l, int m, int n, Tgt16 tgt16)
{
    static_assert(K == 8 || K == 16, "a strip is 8 or 16 columns");
    const int NEG = -131072;
    const int strips = (n + K - 1) / K;
    int score = 0;
    for (int s = 0; s < strips; s++) {
        const int b0 = s * K;
        const uint32_t tw = tgt16(b0);
        int S[K], T[K];
        MCX_UNROLL
        for (int k = 0; k < K; k++) { S[k] = -2 - (b0 + k + 1); T[k] = NEG; } // row "-1": s[0][j] = -2 - j, t[0][j] = NEG
        int diag_next = b0 == 0 ? 0 : -2 - b0; // s[0][b0]
        const bool more = s + 1 < strips;
        uint32_t qc = 0, qn = 0;
        for (int a = 0; a < m; a++) {
            if ((a & 15) == 0) { qc = mem.get(l.off_q + 2 * (a >> 4)); qn = mem.get(l.off_q + 2 * (a >> 4) + 1); }
            const uint32_t qb = (qc >> (30 - 2 * (a & 15))) & 3u;
            const bool q_is_n = (qn >> (15 - (a & 15))) & 1u;
            // mismatch bits of the strip: field k (top first) non-zero = the bases differ (a query N — code 4 — differs from every genome base)
            uint32_t x = (qb * 0x55555555u) ^ tw;
            x = (x | (x >> 1)) & 0x55555555u;
            if (q_is_n) x = 0x55555555u;
            int Rl, Sl;
            if (b0 == 0) { Rl = NEG; Sl = -2 - (a + 1); } // r[i][0], s[i][0]
            else { const uint32_t e = mem.get(l.off_edge + a); Rl = (int)(int16_t)(e & 0xFFFFu); Sl = (int)(int16_t)(e >> 16); }
            int diag = diag_next;
            diag_next = Sl; // s[i][b0] is the next row's upper-left neighbour
            uint32_t flags = 0;
            MCX_UNROLL
            for (int k = 0; k < K; k++) {
                const int rr = (Rl - 1 > Sl - 3) ? Rl - 1 : Sl - 3;
                const int tt = (T[k] - 1 > S[k] - 3) ? T[k] - 1 : S[k] - 3;
                const int dg = diag + (((x >> (30 - 2 * k)) & 1u) ? -2 : 2);
                int sc = dg > rr ? dg : rr;
                sc = sc > tt ? sc : tt;
                flags |= ((sc == rr ? 1u : 0u) | (sc == tt ? 2u : 0u)) << (2 * k);
                diag = S[k];
                S[k] = sc; T[k] = tt; Rl = rr; Sl = sc;
            }
            mem.put(l.off_dir + (uint32_t)(s * (int)l.rows + a) * LaneDir<K, true>::words, flags);
            if (more) mem.put(l.off_edge + a, ((uint32_t)Rl & 0xFFFFu) | ((uint32_t)Sl << 16));
        }
        if (!more) {
            const int kk = (n - 1) - b0;
            MCX_UNROLL
            for (int k = 0; k < K; k++) if (k == kk) score = S[k];
        }
    }
    return score;
}

// "query base qi differs from target base tj" for the walks' mismatch counts, with the words that hold them kept at hand
template <class Tgt16>
struct LaneBases {
    const LaneMem &mem; const LaneLayout &l; Tgt16 tgt16;
    int cq = -1, ct = -1;
    uint32_t qc = 0, qn = 0, tw = 0;
    MCX_HD LaneBases(const LaneMem &m, const LaneLayout &lay, Tgt16 t) : mem(m), l(lay), tgt16(t) {}
    MCX_HD int differ(int qi, int tj)
    {
        if ((qi >> 4) != cq) { cq = qi >> 4; qc = mem.get(l.off_q + 2 * cq); qn = mem.get(l.off_q + 2 * cq + 1); }
        if ((tj >> 4) != ct) { ct = tj >> 4; tw = tgt16(ct * 16); }
        const uint32_t qb = (qc >> (30 - 2 * (qi & 15))) & 3u, tb = (tw >> (30 - 2 * (tj & 15))) & 3u;
        return (((qn >> (15 - (qi & 15))) & 1u) || qb != tb) ? 1 : 0;
    }
};

// The walks hand every alignment column — from the last to the first — to a sink: col(kind, differ) with kind 0 'M', 1 'I', 2 'D'
// (the CIGAR codes) and, for 'M', whether the two bases differ (asked for only when sink.wants_bases()).
struct OpsSink { // the pipeline's: the column string back to front into the pair's ops area + the DpSummary
    uint8_t *ops; int w; DpSumAcc acc; bool bases;
    MCX_HD bool wants_bases() const { return bases; }
    MCX_HD void col(int kind, int differ) { ops[--w] = kind == 0 ? 'M' : (kind == 1 ? 'I' : 'D'); acc.put(kind, differ); }
};

// the traceback of nw_alignment (nw_alignment.cpp:59-74) by the lane that swept the problem
template <int K, class Tgt16, class Sink>
static inline MCX_HD void lane_walk_nw(const LaneMem &mem, const LaneLayout &l, int m, int n, Tgt16 tgt16, Sink &sink)
{
    int i = m, j = n; // 1-based matrix indices
    LaneBases<Tgt16> bases(mem, l, tgt16);
    while (i > 0 || j > 0) {
        unsigned d;
        if (i == 0) d = 1;       // s[0][j] == r[0][j]
        else if (j == 0) d = 2;  // s[i][0] == t[i][0]
        else {
            const int a = i - 1, b = j - 1;
            d = (mem.get(l.off_dir + (uint32_t)((b / K) * (int)l.rows + a) * LaneDir<K, true>::words) >> (2 * (b % K))) & 3u;
        }
        if (d & 1) { sink.col(2, 0); j--; }       // '-' inserted into s1 (read string)
        else if (d & 2) { sink.col(1, 0); i--; }  // '-' inserted into s2 (genome string)
        else { sink.col(0, sink.wants_bases() ? bases.differ(i - 1, j - 1) : 0); i--; j--; }
    }
}

// ops: area of m + n bytes; the column string is written back to front; returns its start offset in ops.
template <int K, class Tgt16>
static inline MCX_HD int lane_trace_nw(const LaneMem &mem, const LaneLayout &l, int m, int n, Tgt16 tgt16, uint8_t *ops, DpSummary *sum, uint32_t ops_base)
{
    OpsSink sink; sink.ops = ops; sink.w = m + n; sink.acc.begin(sum); sink.bases = sum != nullptr;
    lane_walk_nw<K>(mem, l, m, n, tgt16, sink);
    sink.acc.end(ops_base + (uint32_t)sink.w, m + n - sink.w);
    return sink.w;
}

// ---------------------------------------------------------------------------------------------
// ksw2 (m=5 q=2 e=1): the difference recurrence; per cell 4 bits: state (0 'M' / 1 / 2) and the two extension bits.
// ---------------------------------------------------------------------------------------------
template <int K, class Tgt16>
static inline MCX_HD void lane_sweep_ksw2(const LaneMem &mem, const LaneLayout &l, int qlen, int tlen, Tgt16 tgt16)
{
    static_assert(K == 8 || K == 16, "a strip is 8 or 16 columns");
    const int Q = 2, QE2 = 6, MAX_SC = 7;
    const int strips = (tlen + K - 1) / K;
    constexpr int DW = LaneDir<K, false>::words;
    for (int s = 0; s < strips; s++) {
        const int b0 = s * K;
        const uint32_t tw = tgt16(b0);
        int U[K], Y[K];
        MCX_UNROLL
        for (int k = 0; k < K; k++) { U[k] = (b0 + k) ? Q : 0; Y[k] = 0; } // the first matrix row (ksw2_alignment.cpp:165)
        const bool more = s + 1 < strips;
        uint32_t qc = 0, qn = 0;
        for (int a = 0; a < qlen; a++) {
            if ((a & 15) == 0) { qc = mem.get(l.off_q + 2 * (a >> 4)); qn = mem.get(l.off_q + 2 * (a >> 4) + 1); }
            const uint32_t qb = (qc >> (30 - 2 * (a & 15))) & 3u;
            const bool q_is_n = (qn >> (15 - (a & 15))) & 1u;
            uint32_t x = (qb * 0x55555555u) ^ tw;
            x = (x | (x >> 1)) & 0x55555555u; // field k: 1 = mismatch
            int xl, vl;
            if (b0 == 0) { xl = 0; vl = a ? Q : 0; } // values entering column 0 (:163)
            else { const uint32_t e = mem.get(l.off_edge + a); xl = (int)(e & 0xFFu); vl = (int)((e >> 8) & 0xFFu) - 8; }
            uint32_t w0 = 0, w1 = 0;
            MCX_UNROLL
            for (int k = 0; k < K; k++) {
                const int sc = q_is_n ? 0 : (((x >> (30 - 2 * k)) & 1u) ? -1 : 1);
                int z = sc + QE2;
                int av = xl + vl;
                const int ut = U[k];
                int bv = Y[k] + ut;
                uint32_t d = av > z ? 1u : 0u;            // signed (:187)
                z = z > av ? z : av;                      // signed max (:188)
                if (bv > z) d = 2u;                       // signed (:189)
                unsigned zu = (unsigned)z & 0xFFu, bu = (unsigned)bv & 0xFFu; // unsigned max / min on bytes (:89-90)
                zu = zu > bu ? zu : bu;
                zu = zu < (unsigned)MAX_SC ? zu : (unsigned)MAX_SC;
                z = (int)zu;
                const int un = z - vl, vn = z - ut;
                z -= Q; av -= z; bv -= z;
                int xn = 0, yn = 0;
                if (av > 0) { xn = av; d |= 4u; }
                if (bv > 0) { yn = bv; d |= 8u; }
                if (DW == 1 || k < 8) w0 |= d << (4 * (k & 7)); else w1 |= d << (4 * (k & 7));
                U[k] = un; Y[k] = yn; xl = xn; vl = vn;
            }
            const uint32_t at = l.off_dir + (uint32_t)(s * (int)l.rows + a) * DW;
            mem.put(at, w0);
            if (DW == 2) mem.put(at + 1, w1);
            if (more) mem.put(l.off_edge + a, (uint32_t)xl | ((uint32_t)(vl + 8) << 8));
        }
    }
}

// ksw_backtrack (ksw2_alignment.cpp:25-68), full band (force_state never fires); i: target index, j: query index
template <int K, class Tgt16, class Sink>
static inline MCX_HD void lane_walk_ksw2(const LaneMem &mem, const LaneLayout &l, int qlen, int tlen, Tgt16 tgt16, Sink &sink)
{
    constexpr int DW = LaneDir<K, false>::words;
    int i = tlen - 1, j = qlen - 1, state = 0;
    LaneBases<Tgt16> bases(mem, l, tgt16);
    while (i >= 0 && j >= 0) {
        const int k = i % K;
        const uint32_t word = mem.get(l.off_dir + (uint32_t)((i / K) * (int)l.rows + j) * DW + (DW == 2 ? (uint32_t)(k >> 3) : 0u));
        const unsigned nib = (word >> (4 * (k & 7))) & 15u;
        const unsigned d = (nib & 3u) | ((nib & 12u) << 1); // the reference's byte: state in bits 0-2, extension bits 3 and 4
        if (state == 0) state = d & 7;
        else if (!((d >> (state + 2)) & 1)) state = 0;
        if (state == 0) state = d & 7;
        if (state == 0) { sink.col(0, sink.wants_bases() ? bases.differ(j, i) : 0); --i; --j; }
        else if (state == 1 || state == 3) { sink.col(2, 0); --i; }
        else { sink.col(1, 0); --j; }
    }
    for (; i >= 0; --i) sink.col(2, 0);
    for (; j >= 0; --j) sink.col(1, 0);
}

template <int K, class Tgt16>
static inline MCX_HD int lane_trace_ksw2(const LaneMem &mem, const LaneLayout &l, int qlen, int tlen, Tgt16 tgt16, uint8_t *ops, DpSummary *sum, uint32_t ops_base)
{
    OpsSink sink; sink.ops = ops; sink.w = qlen + tlen; sink.acc.begin(sum); sink.bases = sum != nullptr;
    lane_walk_ksw2<K>(mem, l, qlen, tlen, tgt16, sink);
    sink.acc.end(ops_base + (uint32_t)sink.w, qlen + tlen - sink.w);
    return sink.w;
}

// 16 codes of an oriented read without N from position p on (2-bit words as k_pack_reads leaves them), first in the top bits;
// p may be negative (down to -15: zeros in front) and may run past the end (the read's words are followed by more of the slice)
static inline MCX_HD uint32_t lane_read16(const uint32_t *codes, int p)
{
    const int w = p >> 4, sh = (p & 15) * 2;
    const uint32_t hi = w >= 0 ? codes[w] : 0u;
    if (!sh) return hi;
    return (hi << sh) | (codes[w + 1] >> (32 - sh));
}

static inline MCX_HD uint32_t lane_reverse16(uint32_t v) // the 16 two-bit symbols of a word in reverse order
{
    v = ((v & 0x00FF00FFu) << 8) | ((v >> 8) & 0x00FF00FFu);
    v = (v << 16) | (v >> 16);
    v = ((v & 0x0F0F0F0Fu) << 4) | ((v >> 4) & 0x0F0F0F0Fu);
    return ((v & 0x33333333u) << 2) | ((v >> 2) & 0x33333333u);
}

// The two strings of a pipeline problem as the sweeps take them.  q = read fragment, t = genome fragment; both reversed on the
// reverse strand (the reference also complements both, which no comparison can see).
// 16 codes of the query from its position p on, out of the read's 2-bit words (a read without N)
static inline MCX_HD uint32_t lane_query16(const uint32_t *codes, int rPos, int qlen, bool rev, int p)
{
    return rev ? lane_reverse16(lane_read16(codes, rPos + qlen - 16 - p)) : lane_read16(codes, rPos + p);
}
// 16 codes of the target from its position b0 on.  Forward strand as it lies; reverse strand t[i] = T[gPos + gLen - 1 - i] =
// 3 - X[f0 + i] with f0 = 2G - gPos - gLen, i.e. the mirrored forward stretch read forwards, complemented
static inline MCX_HD uint32_t lane_target16(const IndexView &ix, int64_t gPos, int tlen, bool rev, int b0)
{
    if (!rev) return ref_codes16(ix, gPos + b0);
    const int64_t f0 = ix.G2 - gPos - (int64_t)tlen;
    if (f0 + b0 + 16 <= ix.G) return ~ref_codes16_fwd(ix, f0 + b0);
    uint32_t v = 0;
    for (int k = 0; k < 16; k++) v = (v << 2) | (b0 + k < tlen ? (uint32_t)ref_code(ix, gPos + tlen - 1 - (b0 + k)) : 0u);
    return v;
}

// One DP problem of the batch pipeline, start to finish, by its lane: stage the query, sweep, walk, hand the column string to the
// fragment (what dp_run_job does with a wavefront).  l: the group's layout (the same in every lane of the wave); mem: this lane's words.
// q = read fragment, t = genome fragment; both reversed on the reverse strand (the reference also complements both, which no
// comparison can see).  Returns the sweep's score (nw: s[m][n] doubled; ksw2: 0 — the pipeline never reads it).
template <int K, bool NW>
static inline MCX_HD int lane_dp_job(const Ctx &cx, const LaneMem &mem, const LaneLayout &l, const DpJob &job, const ReadRef &rd)
{
    const IndexView &ix = cx.ix;
    const int qlen = job.rLen, tlen = job.gLen;
    const bool rev = job.rev != 0;
    lane_stage_query(mem, l, qlen, [&](int p, uint32_t &codes, uint32_t &flags) {
        if (rd.codes) { flags = 0; codes = lane_query16(rd.codes, job.rPos, qlen, rev, p); return; }
        codes = 0; flags = 0;
        for (int k = 0; k < 16 && p + k < qlen; k++) {
            const int c = read_code(rd, rev ? job.rPos + qlen - 1 - (p + k) : job.rPos + p + k);
            codes |= (uint32_t)(c & 3) << (30 - 2 * k);
            flags |= (uint32_t)(c > 3) << (15 - k);
        }
    });
    auto tgt16 = [&](int b0) -> uint32_t { return lane_target16(ix, job.gPos, tlen, rev, b0); };
    PairState st = pair_state(cx.state, cx.lay, cx.caps, job.pair);
    DpSummary *sum = cx.dp_summary ? (DpSummary *)(st.ops + job.ops_off - kDpSum) : nullptr; // (stage_build left room for it)
    int score = 0, w;
    if (NW) {
        score = lane_sweep_nw<K>(mem, l, qlen, tlen, tgt16);
        w = lane_trace_nw<K>(mem, l, qlen, tlen, tgt16, st.ops + job.ops_off, sum, (uint32_t)job.ops_off);
    } else {
        lane_sweep_ksw2<K>(mem, l, qlen, tlen, tgt16);
        w = lane_trace_ksw2<K>(mem, l, qlen, tlen, tgt16, st.ops + job.ops_off, sum, (uint32_t)job.ops_off);
    }
    Frag f = st.frags[job.frag]; // one fetch, one store (the fields share two words)
    f.ops_off = job.ops_off + w;
    f.ops_len = qlen + tlen - w;
    f.meta = sum ? (uint32_t)((job.ops_off - kDpSum) >> 3) + 1u : 0u;
    st.frags[job.frag] = f;
    return score;
}

} // namespace mcx
#endif
