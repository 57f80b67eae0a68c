// mapcaller_amd/csrc/mcx_dp_lane2.h — gapped extension, TWO PROBLEMS PER LANE in 16-bit halves.
//
// The same recurrences, flags and walks as mcx_dp_lane.h (ksw_extz2_sse + ksw_backtrack, reference
// src/ksw2_alignment.cpp:25-248 — whose core is 16 int8 lanes per instruction, macros :74-95; nw_alignment,
// src/nw_alignment.cpp:18-83), with the arithmetic the reference's own vectors have: every value of both
// recurrences fits sixteen bits (ksw2's differences lie in [-8, 14]; nw's doubled scores are bounded by
// 3 (m + n) + 4, far below 2^15 for any problem the lists hold), so a 32-bit register carries the same cell
// of TWO problems — problem A in the low half, problem B in the high half — and one v_pk_*_i16 / _u16
// instruction of gfx950 advances both.  Nothing crosses halves: an instruction's two results never meet.
//
//  * the strip / row order, the strip's right edge parked per row, the lane-interleaved words are
//    mcx_dp_lane.h's; what a lane keeps is laid out for both problems (LaneLayout2);
//  * a comparison becomes arithmetic, because gfx9 has no packed compare: "a > b" = min_u16(max_i16(a - b, 0), 1),
//    "max moved" = min_u16(new - old, 1); the traceback bits of a row are shifted into 16-bit accumulators
//    (acc = acc * 2 + bit, one v_pk_mad_u16) and permuted into one (nw) / two (ksw2) words per problem at
//    the row's end, so that a walk step costs what it cost before;
//  * nw runs on s~[i][j] = s[i][j] - (i + j): a diagonal step then adds 0 or -4 (one multiply-add from the
//    mismatch bit), the two gap steps add -2 / -4; equalities between the three candidates of a cell — all the
//    traceback asks — are untouched by a shift common to the three, and the sentinel for "no gap state yet"
//    becomes one that fits sixteen bits and still loses every maximum (kNeg2);
//  * ksw2's unsigned-byte max / min (ksw2_alignment.cpp:89-90) is the unsigned 16-bit max / min here: a negative
//    sum is large in either width and the minimum with 7 follows, a non-negative one is the same number.
//
// A half whose problem is shorter or narrower than its neighbour's computes cells nobody reads (rows past its
// query, columns past its target, strips past its last): every cell depends on its left, upper and upper-left
// neighbours only, so nothing flows back, and 16-bit wrap-around in such cells is harmless.
//
// The host build (tests/hostemu: the reference's 620 function-level vectors, the golden SAM sets) runs the same
// code with the packed instructions spelt out in C.
#ifndef MCX_DP_LANE2_H
#define MCX_DP_LANE2_H
#include "mcx_dp_lane.h"

#if defined(__HIPCC__)
#define MCX_HDI __host__ __device__ __forceinline__
#else
#define MCX_HDI inline
#endif

namespace mcx {

// ---- the packed 16-bit instructions this file is written in ------------------------------------------------------------------
// On the device: clang's two-element vectors and element-wise builtins, which select v_pk_add_u16 / v_pk_sub_i16 / v_pk_max_i16 / v_pk_max_u16 /
// v_pk_min_u16 / v_pk_mad_u16 one for one — PROVIDED the optimiser cannot see the constants 1, 2 and -4: left to itself it turns min(x, 1) into two
// compares, two selects and a v_perm_b32, and x * 2 + b into a shift and an or.  Those constants therefore come out of opaque() (an empty asm: no
// instruction, but nothing is known about the value afterwards), once per sweep, and sit in registers.  (Inline assembly for the instructions themselves
// was the first form: the hazard recogniser then assumes a 16-bit destination-select write behind every one and puts an s_nop between dependent pairs —
// 154 per row of sixteen cells.)
namespace pk {
static inline MCX_HD uint32_t join(uint32_t lo, uint32_t hi) { return (lo & 0xFFFFu) | (hi << 16); }
static inline MCX_HD uint32_t dup(int v) { return join((uint32_t)v, (uint32_t)v); }
static inline MCX_HD int lo(uint32_t v) { return (int)(int16_t)(v & 0xFFFFu); }
static inline MCX_HD int hi(uint32_t v) { return (int)(int16_t)(v >> 16); }
#if defined(__HIP_DEVICE_COMPILE__)
typedef short s2v __attribute__((ext_vector_type(2)));
typedef unsigned short u2v __attribute__((ext_vector_type(2)));
#define MCX_AS_S(x) __builtin_bit_cast(::mcx::pk::s2v, (uint32_t)(x))
#define MCX_AS_U(x) __builtin_bit_cast(::mcx::pk::u2v, (uint32_t)(x))
#define MCX_BC(x) __builtin_bit_cast(uint32_t, (x))
static __device__ __forceinline__ uint32_t opaque(uint32_t c) { asm("" : "+v"(c)); return c; }
static __device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) { return MCX_BC(MCX_AS_U(a) + MCX_AS_U(b)); }
static __device__ __forceinline__ uint32_t sub(uint32_t a, uint32_t b) { return MCX_BC(MCX_AS_U(a) - MCX_AS_U(b)); }
static __device__ __forceinline__ uint32_t max_i(uint32_t a, uint32_t b) { return MCX_BC(__builtin_elementwise_max(MCX_AS_S(a), MCX_AS_S(b))); }
static __device__ __forceinline__ uint32_t max_u(uint32_t a, uint32_t b) { return MCX_BC(__builtin_elementwise_max(MCX_AS_U(a), MCX_AS_U(b))); }
static __device__ __forceinline__ uint32_t min_u(uint32_t a, uint32_t b) { return MCX_BC(__builtin_elementwise_min(MCX_AS_U(a), MCX_AS_U(b))); }
// a * b + c (the low sixteen bits of each half: the same for signed and unsigned)
static __device__ __forceinline__ uint32_t mad(uint32_t a, uint32_t b, uint32_t c) { return MCX_BC(MCX_AS_U(a) * MCX_AS_U(b) + MCX_AS_U(c)); }
// either half two bits up; two bits down with its sign (together: a half's low fourteen bits as a signed number)
static __device__ __forceinline__ uint32_t shl2(uint32_t a) { return MCX_BC(MCX_AS_U(a) << (::mcx::pk::u2v)(2)); }
static __device__ __forceinline__ uint32_t sar2(uint32_t a) { return MCX_BC(MCX_AS_S(a) >> (::mcx::pk::s2v)(2)); }
#undef MCX_AS_S
#undef MCX_AS_U
#undef MCX_BC
#else
static inline uint32_t opaque(uint32_t c) { return c; }
template <class F> static inline uint32_t each(uint32_t a, uint32_t b, F f) { return join((uint32_t)f(lo(a), lo(b)), (uint32_t)f(hi(a), hi(b))); }
static inline uint32_t add(uint32_t a, uint32_t b) { return each(a, b, [](int x, int y) { return x + y; }); }
static inline uint32_t sub(uint32_t a, uint32_t b) { return each(a, b, [](int x, int y) { return x - y; }); }
static inline uint32_t max_i(uint32_t a, uint32_t b) { return each(a, b, [](int x, int y) { return x > y ? x : y; }); }
static inline uint32_t max_u(uint32_t a, uint32_t b) { return each(a, b, [](int x, int y) { const unsigned p = (unsigned)x & 0xFFFFu, q = (unsigned)y & 0xFFFFu; return (int)(p > q ? p : q); }); }
static inline uint32_t min_u(uint32_t a, uint32_t b) { return each(a, b, [](int x, int y) { const unsigned p = (unsigned)x & 0xFFFFu, q = (unsigned)y & 0xFFFFu; return (int)(p < q ? p : q); }); }
static inline uint32_t mad(uint32_t a, uint32_t b, uint32_t c) { return join((uint32_t)(lo(a) * lo(b) + lo(c)), (uint32_t)(hi(a) * hi(b) + hi(c))); }
static inline uint32_t shl2(uint32_t a) { return join((uint32_t)lo(a) << 2, (uint32_t)hi(a) << 2); }
static inline uint32_t sar2(uint32_t a) { return join((uint32_t)(lo(a) >> 2), (uint32_t)(hi(a) >> 2)); }
#endif
template <int C> static inline MCX_HD uint32_t addc(uint32_t a) { return add(a, dup(C)); } // (an inline constant of the instruction)
} // namespace pk

constexpr int kNeg2 = -16000; // "no gap state yet" in sixteen bits: below every score of a problem with m + n <= 4000, and 2 (n + 1) above -2^15

// where a lane keeps the words of its two problems (the same offsets in every lane of the wave: the group's longest query and widest target)
struct LaneLayout2 {
    uint32_t off_q;    // one word per row: the two query codes of the row (0..3, 4 = N), A low / B high
    uint32_t off_edge; // one word per row: the strip's right edge of both problems, A low / B high (nw: s~ in fourteen bits under min(s - r, 2); ksw2: x | (v + 8) << 8)
    uint32_t off_dir;  // traceback words: ((strip * rows + row) * DW + problem * DW / 2 ...)
    uint32_t off_t;    // two words per strip: the sixteen target codes the strip began with (A, B), kept for the walks' mismatch counts
    uint32_t rows;     // row pitch (the group's longest query)
    uint32_t words;    // total
};
template <bool NW> struct LaneDir2 { static constexpr int per_problem = NW ? 1 : 2, words = 2 * per_problem; };

template <int K, bool NW>
static inline MCX_HD LaneLayout2 lane_layout2(int rows, int strips)
{
    LaneLayout2 l;
    l.rows = (uint32_t)rows;
    l.off_q = 0;
    l.off_edge = (uint32_t)rows;
    l.off_dir = l.off_edge + (uint32_t)rows;
    l.off_t = l.off_dir + (uint32_t)strips * (uint32_t)rows * (uint32_t)LaneDir2<NW>::words;
    l.words = l.off_t + 2u * (uint32_t)strips;
    return l;
}

// the two queries into the lane's words.  code(h, p): the code (0..3, 4 for N) of problem h's query position p (asked for p < its length only)
template <class Code>
static inline MCX_HD void lane_stage_query2(const LaneMem &m, const LaneLayout2 &l, int qlen_a, int qlen_b, Code code)
{
    const int rows = qlen_a > qlen_b ? qlen_a : qlen_b;
    for (int p = 0; p < rows; p++) m.put(l.off_q + (uint32_t)p, pk::join(p < qlen_a ? (uint32_t)code(0, p) : 0u, p < qlen_b ? (uint32_t)code(1, p) : 0u));
}

// sixteen codes of a query (2-bit fields, first in the top bits) + N flags (bit 15 first) -> its rows' halves, sixteen rows at a time
template <class Get16A, class Get16B>
static inline MCX_HD void lane_stage_query2_words(const LaneMem &m, const LaneLayout2 &l, int qlen_a, int qlen_b, Get16A get_a, Get16B get_b)
{
    const int rows = qlen_a > qlen_b ? qlen_a : qlen_b;
    for (int p0 = 0; p0 < rows; p0 += 16) {
        uint32_t ca = 0, na = 0, cb = 0, nb = 0;
        if (p0 < qlen_a) get_a(p0, ca, na);
        if (p0 < qlen_b) get_b(p0, cb, nb);
        const int n = rows - p0 < 16 ? rows - p0 : 16;
        for (int k = 0; k < n; k++) {
            const uint32_t a = ((na >> (15 - k)) & 1u) ? 4u : (ca >> (30 - 2 * k)) & 3u, b = ((nb >> (15 - k)) & 1u) ? 4u : (cb >> (30 - 2 * k)) & 3u;
            m.put(l.off_q + (uint32_t)(p0 + k), a | (b << 16));
        }
    }
}

// the K target codes of a strip, one register per column, A low / B high
template <int K>
static inline MCX_HD void lane_targets2(uint32_t ta, uint32_t tb, uint32_t (&T)[K])
{
    MCX_UNROLL
    for (int k = 0; k < K; k++) T[k] = ((ta >> (30 - 2 * k)) & 3u) | (((tb >> (30 - 2 * k)) & 3u) << 16);
}

// ---------------------------------------------------------------------------------------------
// nw.  tgt_a / tgt_b (b0): the problem's target codes b0 .. b0+15 as 2-bit fields, first in the top bits.
// A row of a strip leaves one word per problem: bit (K-1-k) = "s != r" of column k (0: a 'D' column), bit 16 + (K-1-k) = "s != t".
// Returns the two final scores s[m][n] (doubled, as lane_sweep_nw) in *score_a / *score_b.
// ---------------------------------------------------------------------------------------------
template <int K, class TgtA, class TgtB>
static inline MCX_HD void lane_sweep_nw2(const LaneMem &mem, const LaneLayout2 &l, int m_a, int n_a, int m_b, int n_b, TgtA tgt_a, TgtB tgt_b, int *score_a, int *score_b,
                                         uint32_t ta0, uint32_t tb0) // ta0 / tb0 = tgt_a(0) / tgt_b(0): fetched by the caller together with everything else a problem starts from
{
    static_assert(K == 8 || K == 16, "a strip is 8 or 16 columns");
    const int m = m_a > m_b ? m_a : m_b, n = n_a > n_b ? n_a : n_b;
    const int strips = (n + K - 1) / K;
    const uint32_t NEG = pk::dup(kNeg2);
    const uint32_t ONE = pk::opaque(pk::dup(1)), TWO = pk::opaque(pk::dup(2)), M4 = pk::opaque(pk::dup(-4));
    uint32_t fin = 0; // s~[m][n] of either half, caught where its row and column pass
    uint32_t ta_nx = ta0, tb_nx = tb0;
    for (int s = 0; s < strips; s++) {
        const int b0 = s * K;
        const uint32_t ta = ta_nx, tb = tb_nx;
        if (s + 1 < strips) { ta_nx = tgt_a(b0 + K); tb_nx = tgt_b(b0 + K); } // (the next strip's target codes — a fetch from the genome — under this strip's rows)
        mem.put(l.off_t + 2u * (uint32_t)s, ta); mem.put(l.off_t + 2u * (uint32_t)s + 1u, tb); // (for the walks)
        uint32_t TG[K];
        lane_targets2<K>(ta, tb, TG);
        uint32_t S[K], T[K];
        MCX_UNROLL
        for (int k = 0; k < K; k++) { S[k] = pk::dup(-2 - 2 * (b0 + k + 1)); T[k] = NEG; } // row 0: s~[0][j] = -2 - 2j, t[0][j] = "none"
        uint32_t diag_next = pk::dup(b0 == 0 ? 0 : -2 - 2 * b0); // s~[0][b0]
        const bool more = s + 1 < strips;
        const int ka = (n_a - 1) - b0, kb = (n_b - 1) - b0; // the column of the strip that is the problem's last (outside 0..K-1: not in this strip)
        // (a row's words — its query codes, the edge the strip before left — are fetched while the row BEFORE it is computed: fetched at the row's own
        //  start they are a wait of several hundred cycles per row with nothing of the lane's own to fill it, and the kernel spent 70 % of its wave-cycles
        //  there (rocprofv3 SQ_WAIT_ANY, profiles/round6).  Row a + 1's edge words are still the strip before's when row a runs: a row stores its own at its end.)
        // The edge a strip leaves for the next one, per row and problem, in ONE half-word: what the next column asks of (r, s) is max(r - 1, s - 3) (in s~: r~ - 2, s~ - 4),
        // and wherever s - r >= 2 that is s - 3 whatever r is — so s~ and min(s - r, 2) carry it exactly: s~ in fourteen bits (a cell's s~ is never below -2 (i + j) - 2 —
        // mismatches down the diagonal and one gap — and launch_dp() keeps m + n within 3000) under the two bits of the difference.  Half the edge words of the first form: the DP stage moves 100 GB a step at config 5
        // (rocprofv3 FETCH_SIZE + WRITE_SIZE), most of it these words and the traceback bits on their way out to HBM and back.
        uint32_t q_nx = mem.get(l.off_q), e_nx = 0;
        if (b0 != 0) e_nx = mem.get(l.off_edge);
        for (int a = 0; a < m; a++) {
            const uint32_t q = q_nx;
            uint32_t Rl, Sl;
            if (b0 == 0) { Rl = NEG; Sl = pk::dup(-2 - 2 * (a + 1)); } // r[i][0], s~[i][0]
            else { Sl = pk::sar2(pk::shl2(e_nx)); Rl = pk::sub(Sl, (e_nx >> 14) & 0x00030003u); }
            if (a + 1 < m) {
                q_nx = mem.get(l.off_q + (uint32_t)(a + 1));
                if (b0 != 0) e_nx = mem.get(l.off_edge + (uint32_t)(a + 1));
            }
            uint32_t diag = diag_next;
            diag_next = Sl; // s~[i][b0] is the next row's upper-left neighbour
            uint32_t fr = 0, ft = 0;
            MCX_UNROLL
            for (int k = 0; k < K; k++) {
                const uint32_t rr = pk::max_i(pk::addc<-2>(Rl), pk::addc<-4>(Sl));
                const uint32_t tt = pk::max_i(pk::addc<-2>(T[k]), pk::addc<-4>(S[k]));
                const uint32_t mm = pk::min_u(TG[k] ^ q, ONE);      // 1 = the bases differ (a query N, code 4, differs from every genome base)
                const uint32_t dg = pk::mad(mm, M4, diag);          // +2 / -2 of the doubled scores, less the diagonal's 2
                const uint32_t sc = pk::max_i(pk::max_i(dg, rr), tt);
                fr = pk::mad(fr, TWO, pk::min_u(pk::sub(sc, rr), ONE)); // sc >= rr: the difference is 0 exactly when they are equal
                ft = pk::mad(ft, TWO, pk::min_u(pk::sub(sc, tt), ONE));
                diag = S[k];
                S[k] = sc; T[k] = tt; Rl = rr; Sl = sc;
            }
            const uint32_t at = l.off_dir + (uint32_t)(s * (int)l.rows + a) * LaneDir2<true>::words;
            mem.put(at, (fr & 0xFFFFu) | (ft << 16));
            mem.put(at + 1u, (fr >> 16) | (ft & 0xFFFF0000u));
            if (more) mem.put(l.off_edge + (uint32_t)a, (Sl & 0x3FFF3FFFu) | (pk::min_u(pk::sub(Sl, Rl), TWO) << 14));
            if (a == m_a - 1 && ka >= 0 && ka < K) { MCX_UNROLL for (int k = 0; k < K; k++) if (k == ka) fin = (fin & 0xFFFF0000u) | (S[k] & 0xFFFFu); }
            if (a == m_b - 1 && kb >= 0 && kb < K) { MCX_UNROLL for (int k = 0; k < K; k++) if (k == kb) fin = (fin & 0xFFFFu) | (S[k] & 0xFFFF0000u); }
        }
    }
    *score_a = pk::lo(fin) + m_a + n_a;
    *score_b = pk::hi(fin) + m_b + n_b;
}

// What a walk reads of its problem: a window of 8 x 8 cells — rows r0 .. r0 - 7, columns c0 .. c0 - 7, the cell the walk stands on at its upper end — taken
// in by EVERY lane of the wavefront at the same step, every eighth one.  A step moves up a row, left a column or both, so the eight steps that follow stay
// inside the window and touch no memory but the column string they write.  (The first form fetched when a lane left its window, four and then eight rows of
// one strip: with 128 walks to a wavefront some lane left its window at nearly every step, and the whole wavefront ran the fetch and waited for it each
// time — the walks were 11-13 of config 5's 39 ms DP stage, scripts/gpu_r6_walk.sh, and the larger window changed nothing.)  A row of the window is ONE
// register: the columns may lie in two strips, whose words are put side by side (column k of a strip is bit K-1-k) and cut to the window's eight columns,
// bit t = column c0 - t; nw: "s == r" bits | "s == t" bits << 8; ksw2: a > z | b > z << 8 | x | y extension bits << 16 / << 24.  Rows picked by
// comparison: an array indexed by a run-time row would live in scratch memory.
template <int K, bool NW>
struct LaneWindow2 {
    static constexpr int N = 8;
    const LaneMem &mem; const LaneLayout2 &l; int h;
    int r0 = 0, c0 = 0, s0 = 0;
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0, w7 = 0;
    uint32_t qp = 0, tw = 0, tp = 0; // the rows' query codes of this problem, three bits each (row r0 - k at bit 3 k); the target codes of strip s0 and of the one before
    MCX_HD LaneWindow2(const LaneMem &m, const LaneLayout2 &lay, int half) : mem(m), l(lay), h(half) {}
    // a row's word from the strips' words (prv: the strip before's, or anything where the window does not reach into it: pm = 0 then)
    static MCX_HDI uint32_t cut(uint32_t cur, uint32_t prv, uint32_t pm, int sh)
    {
        prv &= pm;
        return (((((prv & 0xFFFFu) << K) | (cur & 0xFFFFu)) >> sh) & 0xFFu) | ((((((prv >> 16) << K) | (cur >> 16)) >> sh) & 0xFFu) << 8);
    }
    MCX_HDI void fetch(int row, int col)
    {
        constexpr uint32_t W = LaneDir2<NW>::words, P = LaneDir2<NW>::per_problem;
        const int s = col / K, sh = K - 1 - (col % K);
        const bool prev = s > 0 && col - (N - 1) < s * K; // the window reaches into the strip before
        const uint32_t pm = prev ? 0xFFFFFFFFu : 0u;
        r0 = row; c0 = col; s0 = s;
        // every fetch is issued before any of them is used, whether the strip before is needed or not (a fetch under a condition is a branch, and the
        // eight rows' fetches behind eight branches waited for one another: the stage took 61 ms that way instead of 39): without it the same word twice
        const uint32_t base_s = l.off_dir + (uint32_t)(s * (int)l.rows) * W + (uint32_t)h * P, base_p = prev ? base_s - l.rows * W : base_s;
        auto up = [&](int k) -> uint32_t { return (uint32_t)(row > k ? row - k : 0) ; }; // (above row 0: row 0 again, never picked)
        const uint32_t r1 = up(1), r2 = up(2), r3 = up(3), r4 = up(4), r5 = up(5), r6 = up(6), r7 = up(7);
        const uint32_t c_0 = mem.get(base_s + (uint32_t)row * W), c_1 = mem.get(base_s + r1 * W), c_2 = mem.get(base_s + r2 * W), c_3 = mem.get(base_s + r3 * W),
                       c_4 = mem.get(base_s + r4 * W), c_5 = mem.get(base_s + r5 * W), c_6 = mem.get(base_s + r6 * W), c_7 = mem.get(base_s + r7 * W);
        const uint32_t p_0 = mem.get(base_p + (uint32_t)row * W), p_1 = mem.get(base_p + r1 * W), p_2 = mem.get(base_p + r2 * W), p_3 = mem.get(base_p + r3 * W),
                       p_4 = mem.get(base_p + r4 * W), p_5 = mem.get(base_p + r5 * W), p_6 = mem.get(base_p + r6 * W), p_7 = mem.get(base_p + r7 * W);
        const uint32_t q0 = mem.get(l.off_q + (uint32_t)row), q1 = mem.get(l.off_q + r1), q2 = mem.get(l.off_q + r2), q3 = mem.get(l.off_q + r3),
                       q4 = mem.get(l.off_q + r4), q5 = mem.get(l.off_q + r5), q6 = mem.get(l.off_q + r6), q7 = mem.get(l.off_q + r7);
        tw = mem.get(l.off_t + 2u * (uint32_t)s + (uint32_t)h); // the strips' target codes as the sweep fetched them
        tp = mem.get(l.off_t + 2u * (uint32_t)(prev ? s - 1 : s) + (uint32_t)h);
        w0 = cut(c_0, p_0, pm, sh); w1 = cut(c_1, p_1, pm, sh); w2 = cut(c_2, p_2, pm, sh); w3 = cut(c_3, p_3, pm, sh);
        w4 = cut(c_4, p_4, pm, sh); w5 = cut(c_5, p_5, pm, sh); w6 = cut(c_6, p_6, pm, sh); w7 = cut(c_7, p_7, pm, sh);
        if (!NW) { // the extension bits: the second word of a row
            const uint32_t d_0 = mem.get(base_s + (uint32_t)row * W + 1u), d_1 = mem.get(base_s + r1 * W + 1u), d_2 = mem.get(base_s + r2 * W + 1u), d_3 = mem.get(base_s + r3 * W + 1u),
                           d_4 = mem.get(base_s + r4 * W + 1u), d_5 = mem.get(base_s + r5 * W + 1u), d_6 = mem.get(base_s + r6 * W + 1u), d_7 = mem.get(base_s + r7 * W + 1u);
            const uint32_t e_0 = mem.get(base_p + (uint32_t)row * W + 1u), e_1 = mem.get(base_p + r1 * W + 1u), e_2 = mem.get(base_p + r2 * W + 1u), e_3 = mem.get(base_p + r3 * W + 1u),
                           e_4 = mem.get(base_p + r4 * W + 1u), e_5 = mem.get(base_p + r5 * W + 1u), e_6 = mem.get(base_p + r6 * W + 1u), e_7 = mem.get(base_p + r7 * W + 1u);
            w0 |= cut(d_0, e_0, pm, sh) << 16; w1 |= cut(d_1, e_1, pm, sh) << 16; w2 |= cut(d_2, e_2, pm, sh) << 16; w3 |= cut(d_3, e_3, pm, sh) << 16;
            w4 |= cut(d_4, e_4, pm, sh) << 16; w5 |= cut(d_5, e_5, pm, sh) << 16; w6 |= cut(d_6, e_6, pm, sh) << 16; w7 |= cut(d_7, e_7, pm, sh) << 16;
        }
        const int qs = 16 * h;
        qp = ((q0 >> qs) & 7u) | (((q1 >> qs) & 7u) << 3) | (((q2 >> qs) & 7u) << 6) | (((q3 >> qs) & 7u) << 9) | (((q4 >> qs) & 7u) << 12) | (((q5 >> qs) & 7u) << 15)
           | (((q6 >> qs) & 7u) << 18) | (((q7 >> qs) & 7u) << 21);
    }
    // the window's word of a row, shifted so that the cell's own bits are bits 0, 8, 16 and 24
    // (the eight words by value: a choice among the FIELDS becomes a fetch from a chosen address, and the walk's state then lives in scratch memory)
    static MCX_HDI uint32_t pick(int d, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4, uint32_t x5, uint32_t x6, uint32_t x7)
    {
        const uint32_t lo = (d & 2) ? ((d & 1) ? x3 : x2) : ((d & 1) ? x1 : x0), hi = (d & 2) ? ((d & 1) ? x7 : x6) : ((d & 1) ? x5 : x4);
        return (d & 4) ? hi : lo;
    }
    MCX_HDI uint32_t cell(int row, int col) const { return pick(r0 - row, w0, w1, w2, w3, w4, w5, w6, w7) >> (c0 - col); }
    // "the query base of this row differs from the target base of this column": a query N (code 4) differs from every genome base
    MCX_HDI int differ(int row, int col) const
    {
        const uint32_t t = tw ^ ((tw ^ tp) & (col >= s0 * K ? 0u : 0xFFFFFFFFu)); // (not "cond ? tw : tp": a choice between two fields becomes a fetch from a chosen address, and the walk's state then lives in memory)
        return ((qp >> (3 * (r0 - row))) & 7u) != ((t >> (30 - 2 * (col % K))) & 3u) ? 1 : 0;
    }
};

// the column string of a walk, back to front, four columns to a store (a walk is a chain of dependent fetches, and on this chip a fetch waits for the
// stores issued before it as well: a byte per store and column was one more thing every fetch of the chain queued behind)
struct OpsSink2 {
    uint8_t *ops; int w; uint32_t word; DpSumAcc acc; bool bases;
    MCX_HDI void begin(uint8_t *area, int len, DpSummary *sum) { ops = area; w = len; word = 0; acc.begin(sum); bases = sum != nullptr; }
    MCX_HD bool wants_bases() const { return bases; }
    MCX_HDI void col(int kind, int differ)
    {
        --w;
        word |= (uint32_t)(kind == 0 ? 'M' : (kind == 1 ? 'I' : 'D')) << (8 * (w & 3));
        if ((w & 3) == 0) { *(uint32_t *)(ops + w) = word; word = 0; } // (the area starts on an 8-byte boundary of the pair's pool and is rounded up to one: stage_build)
        acc.put(kind, differ);
    }
    MCX_HDI void end(uint32_t ops_base, int len)
    {
        for (int k = w; k & 3; k++) ops[k] = (uint8_t)(word >> (8 * (k & 3))); // the string's first columns, short of a word
        acc.end(ops_base + (uint32_t)w, len - w);
    }
};

// One problem's walk as a state machine, so that a lane's two walks — and the wavefront's 128 — advance together.
// nw: nw_alignment's traceback (nw_alignment.cpp:59-74); ksw2: ksw_backtrack (ksw2_alignment.cpp:25-68), full band; i: target index, j: query index.
template <int K, bool NW>
struct LaneWalk2 {
    LaneWindow2<K, NW> win;
    OpsSink2 sink;
    int i, j, state;
    bool live;
    MCX_HD LaneWalk2(const LaneMem &m, const LaneLayout2 &lay, int half) : win(m, lay, half), i(0), j(0), state(0), live(false) {}
    MCX_HDI void begin(int qlen, int tlen, uint8_t *area, DpSummary *sum)
    {
        sink.begin(area, qlen + tlen, sum);
        if (NW) { i = qlen; j = tlen; live = i > 0 || j > 0; } // 1-based matrix indices
        else { i = tlen - 1; j = qlen - 1; live = true; state = 0; }
    }
    // the window from the cell the next step reads (a walk that has reached the matrix's edge stays on it and reads nothing more)
    // (taken in whether the walk needs it or not — cell (0, 0) then: a fetch under a condition is a branch, and the lane's two windows behind two branches
    //  are two waits where one will do)
    MCX_HDI void fetch()
    {
        const bool need = live && (NW ? (i > 0 && j > 0) : (i >= 0 && j >= 0));
        const int row = NW ? i - 1 : j, col = NW ? j - 1 : i;
        win.fetch(need ? row : 0, need ? col : 0);
    }
    MCX_HDI void step()
    {
        if (!live) return;
        if (NW) {
            unsigned d;
            if (i == 0) d = 1;        // s[0][j] == r[0][j]
            else if (j == 0) d = 2;   // s[i][0] == t[i][0]
            else {
                const uint32_t w = win.cell(i - 1, j - 1);
                d = ((w & 1u) ^ 1u) | ((((w >> 8) & 1u) ^ 1u) << 1);
            }
            if (d & 1) { sink.col(2, 0); j--; }
            else if (d & 2) { sink.col(1, 0); i--; }
            else { sink.col(0, sink.wants_bases() ? win.differ(i - 1, j - 1) : 0); i--; j--; }
            live = i > 0 || j > 0;
        } else {
            if (i >= 0 && j >= 0) {
                const uint32_t w = win.cell(j, i);
                const unsigned st = ((w >> 8) & 1u) ? 2u : (w & 1u);
                const unsigned d = st | (((w >> 16) & 1u) << 3) | (((w >> 24) & 1u) << 4); // the reference's byte: state in bits 0-2, extension bits 3 and 4
                if (state == 0) state = d & 7;
                else if (!((d >> (state + 2)) & 1)) state = 0;
                if (state == 0) state = d & 7;
                if (state == 0) { sink.col(0, sink.wants_bases() ? win.differ(j, i) : 0); --i; --j; }
                else if (state == 1 || state == 3) { sink.col(2, 0); --i; }
                else { sink.col(1, 0); --j; }
            } else if (i >= 0) { sink.col(2, 0); --i; }
            else { sink.col(1, 0); --j; }
            live = i >= 0 || j >= 0;
        }
    }
};

// both walks of a lane (have_b false: one); every eighth step all of them take their windows in
template <int K, bool NW>
static MCX_HDI void lane_walk2(LaneWalk2<K, NW> &wa, LaneWalk2<K, NW> &wb)
{
    for (int n = 0; wa.live || wb.live; n++) {
        if ((n & (LaneWindow2<K, NW>::N - 1)) == 0) { wa.fetch(); wb.fetch(); }
        wa.step(); wb.step();
    }
}

// ---------------------------------------------------------------------------------------------
// ksw2 (m=5 q=2 e=1).  A row of a strip leaves two words per problem: word 0 = "a > z" bits | "b > z" bits << 16 (the state: the
// second wins), word 1 = the x / y extension bits the same way; column k at bit K-1-k.
// ---------------------------------------------------------------------------------------------
template <int K, class TgtA, class TgtB>
static inline MCX_HD void lane_sweep_ksw2_2(const LaneMem &mem, const LaneLayout2 &l, int qlen_a, int tlen_a, int qlen_b, int tlen_b, TgtA tgt_a, TgtB tgt_b, uint32_t ta0, uint32_t tb0)
{
    static_assert(K == 8 || K == 16, "a strip is 8 or 16 columns");
    const int Q = 2;
    const uint32_t ONE = pk::opaque(pk::dup(1)), TWO = pk::opaque(pk::dup(2)), SEVEN = pk::opaque(pk::dup(7));
    const int qlen = qlen_a > qlen_b ? qlen_a : qlen_b, tlen = tlen_a > tlen_b ? tlen_a : tlen_b;
    const int strips = (tlen + K - 1) / K;
    uint32_t ta_nx = ta0, tb_nx = tb0;
    for (int s = 0; s < strips; s++) {
        const int b0 = s * K;
        const uint32_t ta = ta_nx, tb = tb_nx;
        if (s + 1 < strips) { ta_nx = tgt_a(b0 + K); tb_nx = tgt_b(b0 + K); }
        mem.put(l.off_t + 2u * (uint32_t)s, ta); mem.put(l.off_t + 2u * (uint32_t)s + 1u, tb);
        uint32_t TG[K];
        lane_targets2<K>(ta, tb, TG);
        uint32_t U[K], Y[K];
        MCX_UNROLL
        for (int k = 0; k < K; k++) { U[k] = pk::dup((b0 + k) ? Q : 0); Y[k] = 0; } // the first matrix row (ksw2_alignment.cpp:165)
        const bool more = s + 1 < strips;
        uint32_t q_nx = mem.get(l.off_q), e_nx = 0; // (the next row's words under this row's cells: lane_sweep_nw2; the edge of both problems in one word: x | (v + 8) << 8 per half)
        if (b0 != 0) e_nx = mem.get(l.off_edge);
        for (int a = 0; a < qlen; a++) {
            const uint32_t q = q_nx;
            // the row's substitution scores + q + 2e as 7 - 2 mm, or 6 whatever the target where the query holds an N (score 0: :150-158)
            const uint32_t is_n = (q >> 2) & 0x00010001u;
            const uint32_t c1 = pk::add(pk::add(is_n, is_n), pk::dup(-2)), c0 = pk::sub(pk::dup(7), is_n);
            uint32_t xl, vl;
            if (b0 == 0) { xl = 0; vl = pk::dup(a ? Q : 0); } // values entering column 0 (:163)
            else { xl = e_nx & 0x00FF00FFu; vl = pk::sub((e_nx >> 8) & 0x00FF00FFu, pk::dup(8)); }
            if (a + 1 < qlen) {
                q_nx = mem.get(l.off_q + (uint32_t)(a + 1));
                if (b0 != 0) e_nx = mem.get(l.off_edge + (uint32_t)(a + 1));
            }
            uint32_t f1 = 0, f2 = 0, fx = 0, fy = 0;
            MCX_UNROLL
            for (int k = 0; k < K; k++) {
                const uint32_t mm = pk::min_u(TG[k] ^ q, ONE);
                const uint32_t z0 = pk::mad(mm, c1, c0);
                uint32_t av = pk::add(xl, vl);
                const uint32_t ut = U[k];
                uint32_t bv = pk::add(Y[k], ut);
                const uint32_t z1 = pk::max_i(z0, av);                                        // signed max (:188)
                f1 = pk::mad(f1, TWO, pk::min_u(pk::sub(z1, z0), ONE));                        // a > z (:187): the maximum moved
                f2 = pk::mad(f2, TWO, pk::min_u(pk::max_i(pk::sub(bv, z1), 0u), ONE));         // b > z, signed (:189)
                const uint32_t z = pk::min_u(pk::max_u(z1, bv), SEVEN);                        // unsigned max, then min with max_sc (:89-90, :190-191)
                const uint32_t un = pk::sub(z, vl), vn = pk::sub(z, ut);
                const uint32_t zq = pk::addc<-2>(z);
                av = pk::sub(av, zq); bv = pk::sub(bv, zq);
                const uint32_t xn = pk::max_i(av, 0u), yn = pk::max_i(bv, 0u);
                fx = pk::mad(fx, TWO, pk::min_u(xn, ONE));
                fy = pk::mad(fy, TWO, pk::min_u(yn, ONE));
                U[k] = un; Y[k] = yn; xl = xn; vl = vn;
            }
            const uint32_t at = l.off_dir + (uint32_t)(s * (int)l.rows + a) * LaneDir2<false>::words;
            mem.put(at, (f1 & 0xFFFFu) | (f2 << 16));
            mem.put(at + 1u, (fx & 0xFFFFu) | (fy << 16));
            mem.put(at + 2u, (f1 >> 16) | (f2 & 0xFFFF0000u));
            mem.put(at + 3u, (fx >> 16) | (fy & 0xFFFF0000u));
            if (more) mem.put(l.off_edge + (uint32_t)a, (xl & 0x00FF00FFu) | ((pk::add(vl, pk::dup(8)) & 0x00FF00FFu) << 8));
        }
    }
}

// Two DP problems of the batch pipeline, start to finish, by their lane (what lane_dp_job does for one).  have_b false: the lane holds one
// problem (the list's last, odd one).  scores[2]: nw's s[m][n] doubled (ksw2: 0, as before).
template <int K, bool NW>
static inline MCX_HD void lane_dp_job2(const Ctx &cx, const LaneMem &mem, const LaneLayout2 &l, const DpJob &job_a, const ReadRef &rd_a, bool have_b, const DpJob &job_b,
                                       const ReadRef &rd_b, int scores[2])
{
    const IndexView &ix = cx.ix;
    const int qa = job_a.rLen, ta = job_a.gLen, qb = have_b ? job_b.rLen : 0, tb = have_b ? job_b.gLen : 0;
    auto get16 = [&](const DpJob &job, const ReadRef &rd, int p, uint32_t &codes, uint32_t &flags) {
        const bool rev = job.rev != 0;
        if (rd.codes) { flags = 0; codes = lane_query16(rd.codes, job.rPos, job.rLen, rev, p); return; }
        codes = 0; flags = 0;
        for (int k = 0; k < 16 && p + k < job.rLen; k++) {
            const int c = read_code(rd, rev ? job.rPos + job.rLen - 1 - (p + k) : job.rPos + p + k);
            codes |= (uint32_t)(c & 3) << (30 - 2 * k);
            flags |= (uint32_t)(c > 3) << (15 - k);
        }
    };
    auto tgt_a = [&](int b0) -> uint32_t { return b0 < ta ? lane_target16(ix, job_a.gPos, ta, job_a.rev != 0, b0) : 0u; };
    auto tgt_b = [&](int b0) -> uint32_t { return b0 < tb ? lane_target16(ix, job_b.gPos, tb, job_b.rev != 0, b0) : 0u; };
    // everything the two problems start from that lies in HBM, asked for at once, before the first store (a fetch behind a store to memory that may be the
    // same waits for it): the first strip's target codes, the fragments the results go to — the queries' words follow in the staging loop
    PairState st_a = pair_state(cx.state, cx.lay, cx.caps, job_a.pair), st_b = pair_state(cx.state, cx.lay, cx.caps, job_b.pair);
    const uint32_t ta0 = tgt_a(0), tb0 = tgt_b(0);
    Frag fr_a = st_a.frags[job_a.frag], fr_b = st_b.frags[job_b.frag]; // one fetch, one store each (the fields share two words)
    lane_stage_query2_words(mem, l, qa, qb, [&](int p, uint32_t &c, uint32_t &f) { get16(job_a, rd_a, p, c, f); }, [&](int p, uint32_t &c, uint32_t &f) { get16(job_b, rd_b, p, c, f); });
    scores[0] = scores[1] = 0;
    if (NW) lane_sweep_nw2<K>(mem, l, qa, ta, qb, tb, tgt_a, tgt_b, &scores[0], &scores[1], ta0, tb0);
    else lane_sweep_ksw2_2<K>(mem, l, qa, ta, qb, tb, tgt_a, tgt_b, ta0, tb0);
    LaneWalk2<K, NW> wa(mem, l, 0), wb(mem, l, 1);
    DpSummary *sum_a = cx.dp_summary ? (DpSummary *)(st_a.ops + job_a.ops_off - kDpSum) : nullptr; // (stage_build left room for it)
    DpSummary *sum_b = cx.dp_summary ? (DpSummary *)(st_b.ops + job_b.ops_off - kDpSum) : nullptr;
    wa.begin(qa, ta, st_a.ops + job_a.ops_off, sum_a);
    if (have_b) wb.begin(qb, tb, st_b.ops + job_b.ops_off, sum_b);
#ifndef MCX_DBG_SKIP_WALK // (an experiment's build: what the walks cost — the results are not alignments then)
    lane_walk2(wa, wb);
#endif
    {
        wa.sink.end((uint32_t)job_a.ops_off, qa + ta);
        Frag f = fr_a;
        f.ops_off = job_a.ops_off + wa.sink.w;
        f.ops_len = qa + ta - wa.sink.w;
        f.meta = sum_a ? (uint32_t)((job_a.ops_off - kDpSum) >> 3) + 1u : 0u;
        st_a.frags[job_a.frag] = f;
    }
    if (have_b) {
        wb.sink.end((uint32_t)job_b.ops_off, qb + tb);
        Frag f = fr_b;
        f.ops_off = job_b.ops_off + wb.sink.w;
        f.ops_len = qb + tb - wb.sink.w;
        f.meta = sum_b ? (uint32_t)((job_b.ops_off - kDpSum) >> 3) + 1u : 0u;
        st_b.frags[job_b.frag] = f;
    }
}

} // namespace mcx
#endif
