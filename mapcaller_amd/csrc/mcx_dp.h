// mapcaller_amd/csrc/mcx_dp.h — gapped extension kernels for gfx950 (device only).
//
// Replaces ksw_extz2_sse + ksw_backtrack (reference src/ksw2_alignment.cpp:25-248, called with
// m=5 q=2 e=1 w=-1 from ksw2_alignment :260) and nw_alignment (src/nw_alignment.cpp:18-83).
//
// Shape of the work: many tiny independent problems (median 3x3, p99
// ~100x100), integer, no reuse -> not a contraction, no MFMA.  One wavefront owns one problem
// and sweeps its anti-diagonals: lane t holds column t of the matrix (K columns per lane for
// targets longer than 64), the difference-recurrence state (u,v,x,y / R,S,T) lives in VGPRs,
// the left neighbour's values arrive by a one-lane wave shift, the query is read from LDS, and
// the per-cell traceback codes go to an LDS-resident band buffer (spilling to a per-block HBM
// scratch only for the rare large problem).  A single lane then walks the traceback and writes
// the column string ('M','I','D') back to front into the caller's ops area.
#ifndef MCX_DP_H
#define MCX_DP_H
#include "mcx_glue.h"
#include "mcx_dp_lane.h"

namespace mcx {

#if defined(__HIPCC__)

constexpr int kDpLdsSeq = 2 * 1024;    // query + target codes in LDS (1 KB each)
constexpr int kDpLdsDir = 12 * 1024;   // traceback bytes kept in LDS per wave
constexpr int kDpSpillSeq = 4096;      // spill area reserved for sequences (2 KB each)

// value of lane (lane-1) of the W-lane group; lane 0 of the group receives `carry`
template <int W>
static __device__ __forceinline__ int lane_shift_up(int v, int carry, int lane)
{
    if (W == 1) return carry; // a group of one lane: the neighbour is the lane's own previous column
    // data-parallel primitives instead of a trip through the LDS crossbar: a lane without a source keeps `carry`
    if (W == 16) return __builtin_amdgcn_update_dpp(carry, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    const int s = __builtin_amdgcn_update_dpp(carry, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    return (W == 64 || lane != 0) ? s : carry;
}

// value held by lane `src` of the W-lane group
template <int W>
static __device__ __forceinline__ int group_pick(int v, int src)
{
    return W == 1 ? v : __shfl(v, src, W);
}

// Orders the group's traceback stores before the walking lane's loads.  W = 64: the block is one
// wave and a block barrier is exact.  W < 64: several problems share a wave and sit in divergent
// loops, so only a wavefront-scope fence is legal — and sufficient, because a wave's LDS and
// memory operations are issued in order.
template <int W>
static __device__ __forceinline__ void dp_sync()
{
    if (W == 64) __syncthreads();
    else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

struct DpBuf { uint8_t *q, *t, *dir; };

// where a (qlen x tlen) problem keeps its sequences and traceback: LDS when it fits
// (lds holds lds_seq bytes for the two sequences followed by lds_dir bytes of traceback)
static __device__ __forceinline__ DpBuf dp_buffers(int qlen, int tlen, uint8_t *lds, uint8_t *spill, int lds_seq = kDpLdsSeq, int lds_dir = kDpLdsDir)
{
    DpBuf b;
    if (qlen <= lds_seq / 2 && tlen <= lds_seq / 2) { b.q = lds; b.t = lds + lds_seq / 2; }
    else { b.q = spill; b.t = spill + kDpSpillSeq / 2; }
    b.dir = ((int64_t)(qlen + tlen - 1) * tlen <= lds_dir) ? lds + lds_seq : spill + kDpSpillSeq;
    return b;
}

// ---------------------------------------------------------------------------------------------
// ksw2: difference recurrence of ksw_extz2_sse, one int8 lane per column.
// q/t: codes 0..4 (already staged, block synchronised).  ops: area of qlen+tlen bytes; the
// column string is written back to front.  Returns its start offset in ops (all lanes);
// length = qlen + tlen - offset; *score = ez.score.
// ---------------------------------------------------------------------------------------------
// SCORE: ez.score is wanted (the per-call drop-in); the pipeline only uses the column string, and the H bookkeeping
// is a sixth of the work per cell.
// The recurrence's values are differences bounded by the scoring constants (match 1, mismatch -1, q 2, e 1): u, v in
// [-3, 7], x, y in [0, 7], their sums in [-6, 14] — the int8 lanes of the reference never wrap, so plain ints hold the
// same values; only the two unsigned byte operations (:89-90) need the byte view of a negative number.
// ksw_backtrack (:25-68) by one lane; full band: force_state never fires.  Returns the column string's start offset in ops.
static __device__ int dp_ksw2_trace(int qlen, int tlen, const uint8_t *q, const uint8_t *t, const uint8_t *dir, uint8_t *ops, DpSummary *sum, uint32_t ops_base)
{
    int w = qlen + tlen;
    int i = tlen - 1, j = qlen - 1, state = 0;
    DpSumAcc acc; acc.begin(sum);
    while (i >= 0 && j >= 0) {
        const unsigned d = dir[(i + j) * tlen + i];
        if (state == 0) state = d & 7;
        else if (!((d >> (state + 2)) & 1)) state = 0;
        if (state == 0) state = d & 7;
        if (state == 0) { ops[--w] = 'M'; acc.put(0, q[j] != t[i]); --i; --j; }
        else if (state == 1 || state == 3) { ops[--w] = 'D'; acc.put(2, 0); --i; }
        else { ops[--w] = 'I'; acc.put(1, 0); --j; }
    }
    for (; i >= 0; --i) { ops[--w] = 'D'; acc.put(2, 0); }
    for (; j >= 0; --j) { ops[--w] = 'I'; acc.put(1, 0); }
    acc.end(ops_base + (uint32_t)w, qlen + tlen - w);
    return w;
}

// (TRACE false: the sweep only — the traceback bytes stay in dir for a walk of their own, dp_ksw2_trace)
template <int K, int W, bool SCORE, bool TRACE = true>
static __device__ int dp_ksw2_core(int qlen, int tlen, const uint8_t *q, const uint8_t *t, uint8_t *dir, uint8_t *ops,
                                   int *score, DpSummary *sum, uint32_t ops_base)
{
    const int lane = threadIdx.x & (W - 1);
    const int Q = 2, QE = 3, QE2 = 6, MAX_SC = 7; // q, q+e, 2(q+e), mat[0] + 2(q+e)
    int u[K], v[K], x[K], y[K], H[K], tc[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        u[k] = v[k] = x[k] = y[k] = 0; H[k] = -0x40000000;
        const int tt = lane + W * k;
        tc[k] = tt < tlen ? t[tt] : 4;
    }
    const int n_diag = qlen + tlen - 1;
    for (int r = 0; r < n_diag; r++) {
        const int st = r - qlen + 1 > 0 ? r - qlen + 1 : 0, en = r < tlen - 1 ? r : tlen - 1;
        int cx_ = 0, cv_ = r ? Q : 0, cH_ = 0; // values entering column 0 (:163)
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int tt = lane + W * k;
            if (W * k <= en) { // uniform over the group
                const int ox = x[k], ov = v[k], oH = H[k];
                const int xl = lane_shift_up<W>(ox, cx_, lane), vl = lane_shift_up<W>(ov, cv_, lane), Hl = lane_shift_up<W>(oH, cH_, lane);
                cx_ = group_pick<W>(ox, W - 1); cv_ = group_pick<W>(ov, W - 1); cH_ = group_pick<W>(oH, W - 1);
                if (tt == r) { y[k] = 0; u[k] = r ? Q : 0; } // first matrix row (:165)
                if (tt >= st && tt <= en) {
                    const int qb = q[r - tt], tb = tc[k];
                    const int sc = (qb == 4 || tb == 4) ? 0 : (qb == tb ? 1 : -1);
                    int z = sc + QE2;
                    int a = xl + vl;
                    const int ut = u[k];
                    int b = y[k] + ut;
                    int d = a > z ? 1 : 0;                     // signed (:187)
                    z = z > a ? z : a;                         // signed max (:188)
                    if (b > z) d = 2;                          // signed (:189)
                    unsigned zu = (unsigned)z & 0xFFu, bu = (unsigned)b & 0xFFu; // unsigned max / min on bytes (:89-90)
                    zu = zu > bu ? zu : bu;
                    zu = zu < (unsigned)MAX_SC ? zu : (unsigned)MAX_SC;
                    z = (int)zu;                               // (0..7)
                    u[k] = z - vl;
                    v[k] = z - ut;
                    z = z - Q;
                    a = a - z;
                    b = b - z;
                    if (a > 0) { x[k] = a; d |= 0x08; } else x[k] = 0;
                    if (b > 0) { y[k] = b; d |= 0x10; } else y[k] = 0;
                    dir[r * tlen + tt] = (uint8_t)d; // (at most 3071 x 1024: 32-bit offsets)
                    if (SCORE) { // H bookkeeping (:200-239); u8/v8 are unsigned bytes there
                        if (r == 0) H[k] = (int)(uint8_t)v[k] - QE - QE;
                        else if (tt == en) H[k] = en > 0 ? Hl + (int)(uint8_t)u[k] - QE : oH + (int)(uint8_t)v[k] - QE;
                        else H[k] = oH + (int)(uint8_t)v[k] - QE;
                    }
                }
            }
        }
    }
    if (SCORE) { // ez.score = H[tlen-1] after the last diagonal
        const int kk = (tlen - 1) / W, ll = (tlen - 1) & (W - 1);
        int sc = 0;
#pragma unroll
        for (int k = 0; k < K; k++) { const int hv = group_pick<W>(H[k], ll); if (k == kk) sc = hv; }
        *score = sc;
    }
    if (TRACE) {
        dp_sync<W>();
        int w = qlen + tlen;
        if (lane == 0) w = dp_ksw2_trace(qlen, tlen, q, t, dir, ops, sum, ops_base);
        w = group_pick<W>(w, 0);
        dp_sync<W>();
        return w;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// nw: r/t/s recurrence of nw_alignment in doubled integers (all reference scores are multiples
// of 0.5 and exact in float), equality-based traceback.  Rows i = read (q), columns j = genome.
// ---------------------------------------------------------------------------------------------
// the traceback of nw_alignment (nw_alignment.cpp:59-74) by one lane
static __device__ int dp_nw_trace(int m, int n, const uint8_t *q, const uint8_t *t, const uint8_t *dir, uint8_t *ops, DpSummary *sum, uint32_t ops_base)
{
    int w = m + n;
    int i = m, j = n; // 1-based matrix indices
    DpSumAcc acc; acc.begin(sum);
    while (i > 0 || j > 0) {
        unsigned d;
        if (i == 0) d = 1;       // s[0][j] == r[0][j]
        else if (j == 0) d = 2;  // s[i][0] == t[i][0]
        else d = dir[(i + j - 2) * n + (j - 1)];
        if (d & 1) { ops[--w] = 'D'; acc.put(2, 0); j--; }       // '-' inserted into s1 (read string)
        else if (d & 2) { ops[--w] = 'I'; acc.put(1, 0); i--; }  // '-' inserted into s2 (genome string)
        else { ops[--w] = 'M'; acc.put(0, q[i - 1] != t[j - 1]); i--; j--; }
    }
    acc.end(ops_base + (uint32_t)w, m + n - w);
    return w;
}

template <int K, int W, bool TRACE = true>
static __device__ int dp_nw_core(int m, int n, const uint8_t *q, const uint8_t *t, uint8_t *dir, uint8_t *ops, int *score, DpSummary *sum,
                                 uint32_t ops_base)
{
    const int lane = threadIdx.x & (W - 1);
    const int NEG = -131072, EXT = -1, NEW = -3;
    int S[K], T[K], R[K], Sdiag[K], tc[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        S[k] = T[k] = R[k] = 0; Sdiag[k] = 0;
        const int b = lane + W * k;
        tc[k] = b < n ? t[b] : 4;
    }
    const int n_diag = m + n - 1;
    for (int r = 0; r < n_diag; r++) {
        const int st = r - m + 1 > 0 ? r - m + 1 : 0, en = r < n - 1 ? r : n - 1;
        // matrix column j = 0 as seen by column b = 0 at row a = r: R = NEG, S = -2 - (a+1)
        int cR = NEG, cS = -2 - (r + 1);
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int b = lane + W * k;
            if (W * k <= en) {
                const int oR = R[k], oS = S[k], oT = T[k];
                const int Rl = lane_shift_up<W>(oR, cR, lane), Sl = lane_shift_up<W>(oS, cS, lane);
                cR = group_pick<W>(oR, W - 1); cS = group_pick<W>(oS, W - 1);
                if (b >= st && b <= en) {
                    const int a = r - b;
                    // s(a-1, b-1): the left S this lane received one diagonal ago, or the
                    // initialisation row / column of the matrix
                    int sd;
                    if (a == 0) sd = b == 0 ? 0 : -2 - b;
                    else if (b == 0) sd = -2 - a;
                    else sd = Sdiag[k];
                    const int Tu = a == 0 ? NEG : oT, Su = a == 0 ? -2 - (b + 1) : oS;
                    const int rr = (Rl + EXT > Sl + NEW) ? Rl + EXT : Sl + NEW;
                    const int tt = (Tu + EXT > Su + NEW) ? Tu + EXT : Su + NEW;
                    const int dg = sd + (q[a] == tc[k] ? 2 : -2);
                    int s = dg > rr ? dg : rr;
                    s = s > tt ? s : tt;
                    R[k] = rr; T[k] = tt; S[k] = s;
                    dir[r * n + b] = (uint8_t)((s == rr ? 1 : 0) | (s == tt ? 2 : 0));
                }
                Sdiag[k] = Sl; // s(a-1, b-1) of the next diagonal
            }
        }
    }
    {
        const int kk = (n - 1) / W, ll = (n - 1) & (W - 1);
        int sc = 0;
#pragma unroll
        for (int k = 0; k < K; k++) { const int hv = group_pick<W>(S[k], ll); if (k == kk) sc = hv; }
        *score = sc;
    }
    if (TRACE) {
        dp_sync<W>();
        int w = m + n;
        if (lane == 0) w = dp_nw_trace(m, n, q, t, dir, ops, sum, ops_base);
        w = group_pick<W>(w, 0);
        dp_sync<W>();
        return w;
    }
    return 0;
}

template <int K, int W, bool SCORE = false>
static __device__ __forceinline__ int dp_core(bool nw, int qlen, int tlen, const DpBuf &b, uint8_t *ops, int *score, DpSummary *sum, uint32_t ops_base)
{
    return nw ? dp_nw_core<K, W>(qlen, tlen, b.q, b.t, b.dir, ops, score, sum, ops_base) : dp_ksw2_core<K, W, SCORE>(qlen, tlen, b.q, b.t, b.dir, ops, score, sum, ops_base);
}

// the sweep alone (traceback bytes to b.dir) and the walk alone (one lane), for kernels that sweep a group of problems and then
// walk the group's tracebacks one per lane
template <int K, int W>
static __device__ __forceinline__ void dp_sweep(bool nw, int qlen, int tlen, const DpBuf &b, int *score)
{
    if (nw) { (void)dp_nw_core<K, W, false>(qlen, tlen, b.q, b.t, b.dir, nullptr, score, nullptr, 0u); return; }
    (void)dp_ksw2_core<K, W, false, false>(qlen, tlen, b.q, b.t, b.dir, nullptr, score, nullptr, 0u);
}
static __device__ __forceinline__ int dp_trace(bool nw, int qlen, int tlen, const uint8_t *q, const uint8_t *t, const uint8_t *dir, uint8_t *ops, DpSummary *sum, uint32_t ops_base)
{
    return nw ? dp_nw_trace(qlen, tlen, q, t, dir, ops, sum, ops_base) : dp_ksw2_trace(qlen, tlen, q, t, dir, ops, sum, ops_base);
}

// tiny problems (up to 8 x 8, most of the bulk: median 3 x 3): one lane each — the same recurrences with
// K = 8 columns in the lane's registers and a group width of 1, sixty-four problems per wave
constexpr int kDpTiny = 8;
constexpr int kDpTinyLds = 2 * kDpTiny + (2 * kDpTiny - 1) * kDpTiny; // q(8) + t(8) + dir

// small problems (p90 15 x 16): 16 lanes each, four per wave
constexpr int kDpSmallT = 16, kDpSmallQ = 32;
constexpr int kDpSmallLds = 64 + (kDpSmallQ + kDpSmallT - 1) * kDpSmallT; // q(32) + t(16) + pad + dir

// the job lists of the DP stage: dp_class 0..3 + 4: the tiny ones of class 0 (k_dp_tiny), 5: the short ones of class 1 (k_dp_half)
constexpr int kDpClasses = 6;
struct JobSinks { JobSink s[kDpClasses]; uint32_t *unsupported; };
static __device__ __forceinline__ int job_class(const DpJob &j)
{
    const int c = dp_class(j.rLen, j.gLen);
    return (c == 0 && j.rLen <= kDpTiny && j.gLen <= kDpTiny) ? 4 : ((c == 1 && j.gLen <= 32 && j.rLen <= 64) ? 5 : c);
}

#endif // __HIPCC__

} // namespace mcx
#endif
