// mapcaller_amd/csrc/mcx_fast.h — the common case of a read pair in ONE kernel, its state in LDS (device only).
//
// The general path keeps a pair's hits, candidates and fragments in a 5 KB record in HBM that three
// kernels (k_cluster, k_build, k_finish) walk one 16-byte access at a time: every access is a round
// trip to L2 / HBM, and the record makes three of them.  Most pairs need none of that room: a handful
// of seeds per read, one candidate each, a few fragments, no mate rescue, gapped fragments of a few
// dozen cells.  k_pair_fast maps such a pair from its seeds to its output records in one kernel: one
// pair per lane, the lane's seeds / candidates / fragments / 2-bit read words / the stretch of the 2-bit
// genome under each read in a private slice of LDS, the same per-pair logic as the general path
// (mcx_glue.h, over LDS pointers).  Global memory is touched three times: seeds + read words in, genome
// windows in, records out.  A pair with gapped fragments (a third of them) parks its slice in HBM half way,
// the DP kernels of the general path align the fragments of all parked pairs at once, and
// k_pair_fast_finish takes the slices back into LDS for the rest.
//
// A pair that does not fit (more seeds, candidates or fragments than the slice holds, an N in a read, a read
// longer than the slice's words, mate rescue needed, a second live candidate, more than two gapped
// fragments) is left untouched and listed; the general path maps the listed pairs afterwards, from seeding
// on.  Results are those of the general path, pair by pair (tests/test_gpu_parity.py compares both).
#ifndef MCX_FAST_H
#define MCX_FAST_H
#include "mcx_dp.h"

namespace mcx {

#if defined(__HIPCC__)

struct FastCaps {
    int hit_cap;     // seeds per read
    int cand_cap;    // candidates per read
    int slots;       // 16-byte slots of region R: fragments from the bottom, the reads' seeds at the top
    int code_words;  // 2-bit words per read (16 bases each)
    int win_words;   // words of the 2-bit genome per read
    int stride;      // bytes of LDS per lane (16 x an odd number: 16-byte accesses of a wave's lanes then fall into different banks)
    int ends_bytes;  // LDS for the chromosome tables (0: they stay in HBM)
};

static inline FastCaps make_fast_caps(int rlen_max, int n_ends, int n_chr)
{
    FastCaps f;
    f.hit_cap = 4; f.cand_cap = 2;
    f.code_words = (rlen_max + 15) / 16;
    f.win_words = f.code_words + 3;
    int fixed = 2 * f.cand_cap * (int)sizeof(Cand) + 2 * f.code_words * 4 + 2 * f.win_words * 4;
    f.slots = 2 * f.hit_cap + 7;
    f.stride = fixed + f.slots * 16;
    f.stride = (f.stride + 15) / 16 * 16;
    if (((f.stride / 16) & 1) == 0) { f.stride += 16; f.slots++; }
    f.ends_bytes = n_ends <= kLdsEnds ? (n_ends * 12 + n_chr * 8 + 15) / 16 * 16 : 0;
    return f;
}

struct FastIn {
    const Hit *hits;             // [read][hit_cap]: the first seeds of every read (k_seed)
    const uint32_t *packed;      // 2-bit reads (k_pack_reads), wpad words each
    int wpad;
    const uint32_t *read_ext;    // bit 31: the read holds an N
    const uint32_t *read_blocks; // bits 20..31: seeds the read produced
    int32_t est;                 // EstiDistance of the pass
};

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// the stretch [f_lo, f_hi] of the forward genome -> the lane's window; returns the .pac byte the window starts at
static __device__ __forceinline__ int64_t load_window(const IndexView &ix, int64_t f_lo, int64_t f_hi, uint32_t *win)
{
    const int64_t b0 = (f_lo >> 2) & ~(int64_t)3; // a word boundary of the .pac bytes
    const int n_words = (int)(((f_hi >> 2) - b0) / 4) + 1;
    const uint32_t *src = (const uint32_t *)(ix.pac + b0);
    int k = 0;
    for (; k + 4 <= n_words; k += 4) { const u32x4_a4 v = *(const u32x4_a4 *)(src + k); win[k] = v.x; win[k + 1] = v.y; win[k + 2] = v.z; win[k + 3] = v.w; }
    for (; k < n_words; k++) win[k] = src[k];
    return b0;
}

// A pair whose gapped fragments have to be aligned leaves the kernel half way: its slice of LDS goes to HBM as it is,
// followed by what the lane held in registers; the DP kernels of the general path work on the saved slices (fragments at
// offset 0, column strings behind them: the Layout of save_layout), k_pair_fast_finish takes them back into LDS.
struct FastSaved {
    PairHdr h;
    int64_t pac_base[2];
    uint32_t pair;
    int32_t pad[3];
};

static inline MCX_HD int64_t save_stride(const FastCaps &fc) { return (int64_t)fc.stride + (int64_t)sizeof(FastSaved); }

static inline Layout save_layout(const FastCaps &fc)
{
    Layout l;
    l.off_frags = 0; l.off_ops = 0; l.off_jobs = 0;
    l.off_hits = (int64_t)(fc.slots - 2 * fc.hit_cap) * 16; l.off_cands = (int64_t)fc.slots * 16;
    l.stride = save_stride(fc);
    return l;
}

struct FastLane { // where a lane's slice keeps what
    Frag *R; Cand *cand0; uint32_t *codes0, *win0; uint8_t *mine;
};

static __device__ __forceinline__ FastLane fast_lane(uint8_t *lds, const FastCaps &fc, int lane)
{
    FastLane L;
    L.mine = lds + fc.ends_bytes + (size_t)lane * fc.stride;
    L.R = (Frag *)L.mine;
    L.cand0 = (Cand *)(L.mine + fc.slots * 16);
    L.codes0 = (uint32_t *)(L.cand0 + 2 * fc.cand_cap);
    L.win0 = L.codes0 + 2 * fc.code_words;
    return L;
}

// the chromosome tables are searched several times per pair: from LDS
static __device__ __forceinline__ void fast_stage_ends(IndexView &ix, const FastCaps &fc, uint8_t *lds)
{
    if (!fc.ends_bytes) return;
    int64_t *e_pos = (int64_t *)lds;
    int64_t *c_fwd = e_pos + ix.n_ends;
    int32_t *e_chr = (int32_t *)(c_fwd + ix.n_chr);
    for (int i = threadIdx.x; i < ix.n_ends; i += blockDim.x) { e_pos[i] = ix.end_pos[i]; e_chr[i] = ix.end_chr[i]; }
    for (int i = threadIdx.x; i < ix.n_chr; i += blockDim.x) c_fwd[i] = ix.chr_fwd[i];
    __syncthreads();
    ix.end_pos = e_pos; ix.end_chr = e_chr; ix.chr_fwd = c_fwd;
}

// gates, scores, records of a pair whose state sits in the lane's slice (k_finish's work); all lanes of the wave come here
static __device__ __forceinline__ void fast_finish(Ctx &cx, const FastCaps &fc, const ReadBatch &rb, bool go, uint32_t pair, PairState &st, PairHdr &h,
                                                   ReadRef *rd, const IndexView *ixr, AlnRec *recs, PairOut *pout, uint32_t *pool_over)
{
    const int nr = cx.pm.paired ? 2 : 1;
    int n_cig[2] = {0, 0};
    uint8_t *detail2 = nullptr;
    if (go) {
        detail2 = cx.detail ? cx.detail + (int64_t)pair * nr * cx.dlay.stride : nullptr;
        finish_scores(cx, st, rd, (DetailHdr *)detail2, n_cig, ixr);
    }
    const uint32_t want = go ? (uint32_t)(n_cig[0] + n_cig[1]) : 0u;
    const uint32_t at = wave_reserve(cx.cig_pool_n, want);
    if (!go) return;
    const bool fits = at + want <= cx.cig_pool_cap;
    if (!fits) atomicOr(pool_over, 1u);
    const uint32_t off[2] = {at, at + (uint32_t)n_cig[0]};
    finish_records(cx, st, rd, recs + (int64_t)pair * nr, fits ? cx.cig_pool : nullptr, off, n_cig, detail2);
    PairOut o;
    o.flags = h.flags; o.est = h.est; o.est_lo = h.est_lo; o.est_hi = h.est_hi;
    o.pair_dist = h.pair_dist; o.pair_ok = (int16_t)h.pair_ok; o.mapped = (int16_t)h.mapped;
    pout[pair] = o;
    (void)fc; (void)rb;
}

static __device__ __forceinline__ void fast_reads(const Ctx &cx, const ReadBatch &rb, const FastCaps &fc, const FastLane &L, uint32_t pair, ReadRef rd[2])
{
    const int nr = cx.pm.paired ? 2 : 1;
#pragma unroll
    for (int s = 0; s < 2; s++) { // (every index a constant after unrolling: the arrays stay in registers)
        const uint32_t r = pair * nr + (s < nr ? s : 0);
        rd[s].ascii = rb.bases + rb.off[r]; rd[s].rlen = (int)(rb.off[r + 1] - rb.off[r]); rd[s].flipped = (cx.pm.paired && s == 1) ? 1 : 0;
        rd[s].codes = L.codes0 + (s < nr ? s : 0) * fc.code_words;
    }
}

// One pair per lane, from its seeds to its records — or, when gapped fragments have to be aligned, to the saved slice and the
// DP job lists; or, when it does not fit, to the list of the general path.  Dynamic LDS: [chromosome tables][64 lane slices].
__global__ void __launch_bounds__(64) k_pair_fast(Ctx cx, ReadBatch rb, FastIn in, FastCaps fc, uint32_t n_pairs, AlnRec *recs, PairOut *pout,
                                                  uint32_t *spill_ids, uint32_t *n_spill, uint8_t *saved, uint32_t *saved_ids, uint32_t *n_saved, JobSinks sinks,
                                                  uint32_t *cells, uint32_t *pool_over)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x;
    fast_stage_ends(cx.ix, fc, lds);
    const FastLane L = fast_lane(lds, fc, lane);
    Frag *const R = L.R;
    const uint32_t pair = blockIdx.x * 64 + lane;
    const bool active = pair < n_pairs;
    const int nr = cx.pm.paired ? 2 : 1;
    cx.caps.hit_cap = fc.hit_cap; cx.caps.cand_cap = fc.cand_cap; cx.caps.frag_cap = fc.slots;
    bool spill = false;
    PairHdr h;
    PairState st;
    ReadRef rd[2];
    IndexView ixr[2] = {cx.ix, cx.ix}; // per read: pac / pac_base describe the lane's window
    int job_frag[2] = {0, 0}, job_s[2] = {0, 0};
    int n_jobs = 0;
    st.hdr = &h;
    st.hits[0] = (Hit *)(R + (fc.slots - 2 * fc.hit_cap)); st.hits[1] = (Hit *)(R + (fc.slots - fc.hit_cap));
    st.cands[0] = L.cand0; st.cands[1] = L.cand0 + fc.cand_cap;
    st.frags = R; st.ops = L.mine; // (ops offsets are relative to the slice: the strings go behind the fragments)

    if (active) {
        int nh[2] = {0, 0};
        fast_reads(cx, rb, fc, L, pair, rd);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (s >= nr) break;
            const uint32_t r = pair * nr + s;
            nh[s] = (int)(in.read_blocks[r] >> 20);
            if (rd[s].rlen > fc.code_words * 16 || (in.read_ext[r] >> 31) || nh[s] > fc.hit_cap) spill = true;
        }
        if (!spill) {
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (s >= nr) break;
                const uint32_t r = pair * nr + s;
                const Hit *src = in.hits + (uint64_t)r * fc.hit_cap;
                for (int i = 0; i < nh[s]; i++) st.hits[s][i] = src[i];
                const U4 *pk = (const U4 *)(in.packed + (uint64_t)r * in.wpad);
                uint32_t *dst = L.codes0 + s * fc.code_words;
                const int nc = (rd[s].rlen + 15) >> 4;
                for (int k = 0; k < nc; k += 4) {
                    const U4 v = pk[k >> 2];
                    dst[k] = v.x;
                    if (k + 1 < nc) dst[k + 1] = v.y;
                    if (k + 2 < nc) dst[k + 2] = v.z;
                    if (k + 3 < nc) dst[k + 3] = v.w;
                }
            }
            // ---- seeds -> candidates -> pairing (stage_cluster_pair) ----
            h.flags = 0; h.n_frags = 0; h.n_ops = 0; h.pair_dist = 0; h.n_jobs = 0; h.pair_ok = 0; h.mapped = 0; h.pad[0] = h.pad[1] = 0;
            h.n_cands[0] = h.n_cands[1] = 0; h.n_hits[0] = h.n_hits[1] = 0;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                h.sum[s].best = -1; h.sum[s].score = 0; h.sum[s].sub = 0;
                if (s >= nr) break;
                h.n_hits[s] = (int16_t)prep_seeds(st.hits[s], nh[s]);
                int nc = cluster_seeds(cx.ix, cx.pm, rd[s].rlen, st.hits[s], h.n_hits[s], st.cands[s], fc.cand_cap);
                if (nc > fc.cand_cap) { spill = true; nc = 0; }
                h.n_cands[s] = (int16_t)nc;
            }
            h.est = in.est; h.est_lo = 0; h.est_hi = 0x7fffffff; h.n_paired = 0;
            if (!spill && cx.pm.paired) {
                h.n_paired = (int16_t)pair_by_distance(in.est, st.cands[0], h.n_cands[0], st.cands[1], h.n_cands[1], h.est_lo, h.est_hi);
                if (h.n_paired == 0) spill = true; // mate rescue: the general path
            }
        }
        if (!spill) {
            // ---- masking, fragment lists (stage_build up to the classification) ----
            if (cx.pm.paired) mask_unpaired(st.cands[0], h.n_cands[0], st.cands[1], h.n_cands[1]);
            else keep_top_scores(st.cands[0], h.n_cands[0]);
            int live_ci[2] = {-1, -1};
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (s >= nr) break;
                Cand *cs = st.cands[s];
                for (int ci = 0; ci < h.n_cands[s]; ci++) {
                    if (cs[ci].score == 0) { cs[ci].frag_off = (int16_t)h.n_frags; cs[ci].n_frags = 0; continue; }
                    if (live_ci[s] >= 0) { spill = true; break; } // a second live candidate: the general path
                    live_ci[s] = ci;
                    const Cand c = cs[ci];
                    const int room = fc.slots - (s == 0 ? 2 : 1) * fc.hit_cap; // the seeds still to be read sit above
                    if (h.n_frags + 2 * c.count + 2 > room) { spill = true; break; }
                    Frag *f = R + h.n_frags;
                    const int nf = build_frags(cx.ix, rd[s].rlen, st.hits[s] + c.first, c.count, f);
                    cs[ci].frag_off = (int16_t)h.n_frags; cs[ci].n_frags = (int16_t)(nf < 0 ? 0 : nf);
                    if (nf < 0) { cs[ci].score = 0; live_ci[s] = -2; continue; }
                    h.n_frags += nf;
                }
                if (spill) break;
            }
            // ---- the genome under each live candidate -> the lane's windows ----
            if (!spill) {
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    if (s >= nr || live_ci[s] < 0) continue;
                    const Cand &c = st.cands[s][live_ci[s]];
                    const Frag a = R[c.frag_off], b = R[c.frag_off + c.n_frags - 1];
                    const int64_t p0 = a.gPos, p1 = b.gPos + b.gLen; // [p0, p1) in 2G coordinates, one strand (CheckAlignmentValidity)
                    const bool rev = p0 >= cx.ix.G;
                    const int64_t f_lo = rev ? cx.ix.G2 - p1 : p0, f_hi = (rev ? cx.ix.G2 - 1 - p0 : p1 - 1);
                    if (p1 <= p0 || f_lo < 0 || (f_hi >> 2) - ((f_lo >> 2) & ~(int64_t)3) >= (int64_t)fc.win_words * 4 - 4) { spill = true; break; }
                    ixr[s].pac_base = load_window(cx.ix, f_lo, f_hi, L.win0 + s * fc.win_words);
                    ixr[s].pac = (const uint8_t *)(L.win0 + s * fc.win_words);
                }
            }
            // ---- ProcessNormalPair (:155-191): classify each gap fragment; gapped ones become jobs of the DP kernels ----
            if (!spill) {
                h.n_ops = h.n_frags * 16; // the column strings go behind the fragments (the seeds there are spent)
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    if (s >= nr || live_ci[s] < 0) continue;
                    const Cand &c = st.cands[s][live_ci[s]];
                    Frag *f = R + c.frag_off;
                    for (int i = 0; i < c.n_frags; i++) {
                        Frag x = f[i];
                        if (x.kind == kSimple) continue;
                        if (x.rLen > 0 && x.gLen > 0) {
                            bool dp = x.rLen != x.gLen;
                            if (!dp) {
                                const int mm = frag_mismatches(ixr[s], x, rd[s]);
                                dp = mm > 1 && mm >= (int)(x.rLen * 0.2);
                            }
                            if (dp) {
                                if (dp_class(x.rLen, x.gLen) < 0 || n_jobs >= 2 || h.n_ops + x.rLen + x.gLen > fc.slots * 16) { spill = true; break; }
                                x.kind = kDp; x.ops_off = h.n_ops; x.ops_len = 0;
                                if (n_jobs == 0) { job_frag[0] = c.frag_off + i; job_s[0] = s; } else { job_frag[1] = c.frag_off + i; job_s[1] = s; }
                                n_jobs++;
                                h.n_ops += x.rLen + x.gLen;
                            } else { x.kind = kPlain; x.ops_len = x.rLen; }
                        } else if (x.rLen > 0) { x.kind = kIns; x.ops_len = x.rLen; }
                        else { x.kind = kDel; x.ops_len = x.gLen; }
                        f[i] = x;
                    }
                    if (spill) break;
                }
            }
        }
        if (spill) n_jobs = 0;
        h.n_jobs = (int16_t)n_jobs;
    }

    // ---- pairs with gapped fragments: slice and registers to HBM, jobs to the lists of the DP kernels ----
    const bool parked = active && !spill && n_jobs > 0;
    {
        const uint32_t slot = wave_reserve(n_saved, parked ? 1u : 0u);
        uint32_t per_class[kDpClasses] = {0, 0, 0, 0, 0, 0}, my_cells = 0;
        DpJob jb[2];
        int jc[2] = {-1, -1};
        if (parked) {
            uint8_t *dst = saved + (uint64_t)slot * save_stride(fc);
            const U4 *src = (const U4 *)L.mine;
            for (int k = 0; k < fc.stride / 16; k++) ((U4 *)dst)[k] = src[k];
            FastSaved sv; sv.h = h; sv.pac_base[0] = ixr[0].pac_base; sv.pac_base[1] = ixr[1].pac_base; sv.pair = pair; sv.pad[0] = sv.pad[1] = sv.pad[2] = 0;
            *(FastSaved *)(dst + fc.stride) = sv;
            saved_ids[slot] = pair; // (what the DP kernels look the reads up by)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (k >= n_jobs) break;
                const Frag x = R[job_frag[k]];
                DpJob j;
                j.pair = slot; j.slot = (uint16_t)job_s[k]; j.rev = x.gPos >= cx.ix.G ? 1 : 0;
                j.rPos = x.rPos; j.rLen = x.rLen; j.gPos = x.gPos; j.gLen = x.gLen; j.ops_off = x.ops_off; j.frag = job_frag[k]; j.score = 0;
                jb[k] = j; jc[k] = job_class(j);
                my_cells += (uint32_t)(x.rLen * x.gLen);
            }
        }
#pragma unroll
        for (int c = 0; c < kDpClasses; c++) per_class[c] = (jc[0] == c ? 1u : 0u) + (jc[1] == c ? 1u : 0u);
        uint32_t base[kDpClasses];
#pragma unroll
        for (int c = 0; c < kDpClasses; c++) base[c] = wave_reserve(sinks.s[c].count, per_class[c]);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (jc[k] < 0) continue;
#pragma unroll
            for (int c = 0; c < kDpClasses; c++) {
                if (jc[k] != c) continue;
                const uint32_t at = base[c] + (k == 1 && jc[0] == c ? 1u : 0u);
                if (at < sinks.s[c].cap) sinks.s[c].jobs[at] = jb[k];
            }
        }
        for (int o = 32; o > 0; o >>= 1) my_cells += __shfl_down(my_cells, o, 64);
        if (lane == 0 && my_cells) atomicAdd(cells, my_cells);
    }
    const uint32_t sp = wave_reserve(n_spill, (active && spill) ? 1u : 0u);
    if (active && spill) spill_ids[sp] = pair;
    // ---- everything else: gates, scores, records (k_finish) ----
    fast_finish(cx, fc, rb, active && !spill && n_jobs == 0, pair, st, h, rd, ixr, recs, pout, pool_over);
}

// the parked pairs once their gapped fragments are aligned: slice back into LDS, then k_finish's work
__global__ void __launch_bounds__(64) k_pair_fast_finish(Ctx cx, ReadBatch rb, FastCaps fc, const uint8_t *saved, const uint32_t *n_saved, AlnRec *recs,
                                                         PairOut *pout, uint32_t *pool_over)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x;
    const uint32_t n = *n_saved;
    if (blockIdx.x * 64u >= n) return;
    fast_stage_ends(cx.ix, fc, lds);
    const FastLane L = fast_lane(lds, fc, lane);
    const uint32_t slot = blockIdx.x * 64 + lane;
    const bool go = slot < n;
    cx.caps.hit_cap = fc.hit_cap; cx.caps.cand_cap = fc.cand_cap; cx.caps.frag_cap = fc.slots;
    PairHdr h;
    PairState st;
    ReadRef rd[2];
    IndexView ixr[2] = {cx.ix, cx.ix};
    uint32_t pair = 0;
    st.hdr = &h;
    st.hits[0] = (Hit *)(L.R + (fc.slots - 2 * fc.hit_cap)); st.hits[1] = (Hit *)(L.R + (fc.slots - fc.hit_cap));
    st.cands[0] = L.cand0; st.cands[1] = L.cand0 + fc.cand_cap;
    st.frags = L.R; st.ops = L.mine;
    if (go) {
        const uint8_t *src = saved + (uint64_t)slot * save_stride(fc);
        U4 *dst = (U4 *)L.mine;
        for (int k = 0; k < fc.stride / 16; k++) dst[k] = ((const U4 *)src)[k];
        const FastSaved sv = *(const FastSaved *)(src + fc.stride);
        h = sv.h; pair = sv.pair;
#pragma unroll
        for (int s = 0; s < 2; s++) { ixr[s].pac_base = sv.pac_base[s]; ixr[s].pac = (const uint8_t *)(L.win0 + s * fc.win_words); }
        fast_reads(cx, rb, fc, L, pair, rd);
    }
    fast_finish(cx, fc, rb, go, pair, st, h, rd, ixr, recs, pout, pool_over);
}

#endif // __HIPCC__

} // namespace mcx
#endif
