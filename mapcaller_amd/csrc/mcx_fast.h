// mapcaller_amd/csrc/mcx_fast.h — the common case of a read pair in ONE kernel, its state in LDS (device only).
//
// The general path keeps a pair's hits, candidates and fragments in a 5 KB record in HBM that three
// kernels (k_cluster, k_build, k_finish) walk one 16-byte access at a time: every access is a round
// trip to L2 / HBM, and the record makes three of them.  Most pairs need none of that room: a handful
// of seeds per read, one candidate each, a few fragments, no mate rescue, gapped fragments of a few
// dozen cells.  k_pair_fast maps such a pair from its seeds to its output records in one kernel: one
// pair per lane, the lane's seeds / candidates / fragments / 2-bit read words / the stretch of the 2-bit
// genome under each read in a private slice of LDS, the same per-pair logic as the general path
// (mcx_glue.h, over LDS pointers), and the pair's gapped fragments aligned by the whole wavefront in
// between (dp_core<1, 64>, anti-diagonal sweep, band in LDS).  Global memory is touched three times:
// seeds + read words in, genome windows in, records out.
//
// A pair that does not fit (more seeds, candidates or fragments than the slice holds, an N in a read, a read
// longer than the slice's words, mate rescue needed, a second live candidate, a gapped fragment beyond
// 32 x 64) is left untouched and listed; the general path maps the listed pairs afterwards, from seeding
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
    int dp_t, dp_q;  // gapped fragments up to dp_t genome x dp_q read bases are aligned inside the kernel
    int ends_bytes;  // LDS for the chromosome tables (0: they stay in HBM)
    int wave_bytes;  // LDS the wave shares: DP strings and band
};

static inline FastCaps make_fast_caps(int rlen_max, int n_ends, int n_chr)
{
    FastCaps f;
    f.hit_cap = 4; f.cand_cap = 2;
    f.code_words = (rlen_max + 15) / 16;
    f.win_words = f.code_words + 3;
    f.dp_t = 32; f.dp_q = 48;
    int fixed = 2 * f.cand_cap * (int)sizeof(Cand) + 2 * f.code_words * 4 + 2 * f.win_words * 4;
    f.slots = 2 * f.hit_cap + 7;
    f.stride = fixed + f.slots * 16;
    f.stride = (f.stride + 15) / 16 * 16;
    if (((f.stride / 16) & 1) == 0) { f.stride += 16; f.slots++; }
    f.ends_bytes = n_ends <= kLdsEnds ? (n_ends * 12 + n_chr * 8 + 15) / 16 * 16 : 0;
    f.wave_bytes = (f.dp_q + f.dp_t + (f.dp_q + f.dp_t - 1) * f.dp_t + 15) / 16 * 16;
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

struct FastJob { int16_t frag, s, rLen, gLen; int32_t ops_off; };

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

// One pair per lane.  Dynamic LDS: [chromosome tables][the wave's DP strings + band][64 lane slices].
__global__ void __launch_bounds__(64) k_pair_fast(Ctx cx, ReadBatch rb, FastIn in, FastCaps fc, uint32_t n_pairs, AlnRec *recs, PairOut *pout,
                                                  uint32_t *spill_ids, uint32_t *n_spill, uint32_t *pool_over)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x;
    if (fc.ends_bytes) { // the chromosome tables are searched several times per pair: from LDS
        int64_t *e_pos = (int64_t *)lds;
        int64_t *c_fwd = e_pos + cx.ix.n_ends;
        int32_t *e_chr = (int32_t *)(c_fwd + cx.ix.n_chr);
        for (int i = lane; i < cx.ix.n_ends; i += 64) { e_pos[i] = cx.ix.end_pos[i]; e_chr[i] = cx.ix.end_chr[i]; }
        for (int i = lane; i < cx.ix.n_chr; i += 64) c_fwd[i] = cx.ix.chr_fwd[i];
        __syncthreads();
        cx.ix.end_pos = e_pos; cx.ix.end_chr = e_chr; cx.ix.chr_fwd = c_fwd;
    }
    uint8_t *wave = lds + fc.ends_bytes;
    uint8_t *mine = wave + fc.wave_bytes + (size_t)lane * fc.stride;
    Frag *R = (Frag *)mine;
    Cand *cand0 = (Cand *)(mine + fc.slots * 16);
    uint32_t *codes0 = (uint32_t *)(cand0 + 2 * fc.cand_cap);
    uint32_t *win0 = codes0 + 2 * fc.code_words;

    const uint32_t pair = blockIdx.x * 64 + lane;
    const bool active = pair < n_pairs;
    const int nr = cx.pm.paired ? 2 : 1;
    cx.caps.hit_cap = fc.hit_cap; cx.caps.cand_cap = fc.cand_cap; cx.caps.frag_cap = fc.slots;
    bool spill = false;
    PairHdr h;
    PairState st;
    ReadRef rd[2];
    IndexView ixr[2] = {cx.ix, cx.ix}; // per read: pac / pac_base describe the lane's window
    int n_cig[2] = {0, 0};
    FastJob jobs[2];
    int n_jobs = 0;
    uint8_t *detail2 = nullptr;
    st.hdr = &h;
    st.hits[0] = (Hit *)(R + (fc.slots - 2 * fc.hit_cap)); st.hits[1] = (Hit *)(R + (fc.slots - fc.hit_cap));
    st.cands[0] = cand0; st.cands[1] = cand0 + fc.cand_cap;
    st.frags = R; st.ops = mine; // (ops offsets are relative to the slice: the strings go behind the fragments)

    if (active) {
        int nh[2] = {0, 0};
        for (int s = 0; s < nr; s++) {
            const uint32_t r = pair * nr + s;
            rd[s].ascii = rb.bases + rb.off[r]; rd[s].rlen = (int)(rb.off[r + 1] - rb.off[r]); rd[s].flipped = (cx.pm.paired && s == 1) ? 1 : 0;
            rd[s].codes = codes0 + s * fc.code_words;
            nh[s] = (int)(in.read_blocks[r] >> 20);
            if (rd[s].rlen > fc.code_words * 16 || (in.read_ext[r] >> 31) || nh[s] > fc.hit_cap) spill = true;
        }
        if (nr == 1) { rd[1] = rd[0]; }
        if (!spill) {
            for (int s = 0; s < nr; s++) {
                const uint32_t r = pair * nr + s;
                const Hit *src = in.hits + (uint64_t)r * fc.hit_cap;
                for (int i = 0; i < nh[s]; i++) st.hits[s][i] = src[i];
                const U4 *pk = (const U4 *)(in.packed + (uint64_t)r * in.wpad);
                uint32_t *dst = codes0 + s * fc.code_words;
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
                    ixr[s].pac_base = load_window(cx.ix, f_lo, f_hi, win0 + s * fc.win_words);
                    ixr[s].pac = (const uint8_t *)(win0 + s * fc.win_words);
                }
            }
            // ---- ProcessNormalPair (:155-191): classify each gap fragment; gapped ones become jobs of the wave ----
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
                        const bool rev = x.gPos >= cx.ix.G;
                        if (x.rLen > 0 && x.gLen > 0) {
                            bool dp = x.rLen != x.gLen;
                            if (!dp) {
                                int mm = 0;
                                for (int k = 0; k < x.rLen; k++)
                                    if (frag_read_code(x, rd[s], rev, k) != frag_ref_code(ixr[s], x, rev, k)) mm++;
                                dp = mm > 1 && mm >= (int)(x.rLen * 0.2);
                            }
                            if (dp) {
                                if (x.gLen > fc.dp_t || x.rLen > fc.dp_q || n_jobs >= 2 || h.n_ops + x.rLen + x.gLen > fc.slots * 16) { spill = true; break; }
                                x.kind = kDp; x.ops_off = h.n_ops; x.ops_len = 0;
                                FastJob j; j.frag = (int16_t)(c.frag_off + i); j.s = (int16_t)s; j.rLen = (int16_t)x.rLen; j.gLen = (int16_t)x.gLen; j.ops_off = h.n_ops;
                                jobs[n_jobs++] = j;
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
    }

    // ---- the wave aligns its gapped fragments one after the other, all lanes on each (k_dp_sel's shape) ----
    {
        DpBuf b; b.q = wave; b.t = wave + fc.dp_q; b.dir = wave + fc.dp_q + fc.dp_t;
        for (int k = 0; k < 2; k++) {
            uint64_t todo = __ballot(active && k < n_jobs);
            while (todo) {
                const int owner = __builtin_ctzll(todo);
                todo &= todo - 1;
                // the owner's job, its read words and its genome window, read across the wave
                const FastJob mj = jobs[k < n_jobs ? k : 0];
                const int rLen = __shfl((int)mj.rLen, owner, 64), gLen = __shfl((int)mj.gLen, owner, 64), js = __shfl((int)mj.s, owner, 64);
                const int frag = __shfl((int)mj.frag, owner, 64), ops_off = __shfl(mj.ops_off, owner, 64);
                uint8_t *theirs = wave + fc.wave_bytes + (size_t)owner * fc.stride;
                const Frag x = ((const Frag *)theirs)[frag];
                const bool rev = x.gPos >= cx.ix.G;
                ReadRef orr; orr.ascii = nullptr; orr.rlen = 0; orr.flipped = 0;
                orr.codes = (const uint32_t *)(theirs + fc.slots * 16 + 2 * fc.cand_cap * sizeof(Cand)) + js * fc.code_words;
                IndexView oix = cx.ix;
                { // the owner's window of that read
                    const int64_t base = js == 0 ? ixr[0].pac_base : ixr[1].pac_base;
                    const uint32_t blo = __shfl((uint32_t)base, owner, 64), bhi = __shfl((uint32_t)((uint64_t)base >> 32), owner, 64);
                    oix.pac_base = (int64_t)((uint64_t)blo | ((uint64_t)bhi << 32));
                    oix.pac = theirs + fc.slots * 16 + 2 * fc.cand_cap * sizeof(Cand) + 2 * fc.code_words * 4 + (size_t)js * fc.win_words * 4;
                }
                for (int i = lane; i < rLen; i += 64) b.q[i] = (uint8_t)read_code(orr, rev ? x.rPos + rLen - 1 - i : x.rPos + i);
                for (int i = lane; i < gLen; i += 64) b.t[i] = (uint8_t)ref_code(oix, rev ? x.gPos + gLen - 1 - i : x.gPos + i);
                __syncthreads();
                int score = 0;
                const int w = dp_core<1, 64>(cx.pm.use_nw != 0, rLen, gLen, b, theirs + ops_off, &score);
                if (lane == owner) {
                    Frag f = R[frag];
                    f.ops_off = ops_off + w;
                    f.ops_len = rLen + gLen - w;
                    R[frag] = f;
                }
                __syncthreads();
            }
        }
    }

    // ---- gates, scores, records (k_finish) ----
    if (active && !spill) {
        detail2 = cx.detail ? cx.detail + (int64_t)pair * nr * cx.dlay.stride : nullptr;
        h.n_jobs = (int16_t)n_jobs;
        finish_scores(cx, st, rd, (DetailHdr *)detail2, n_cig, ixr);
    }
    const uint32_t want = (active && !spill) ? (uint32_t)(n_cig[0] + n_cig[1]) : 0u;
    const uint32_t at = wave_reserve(cx.cig_pool_n, want);
    const uint32_t sp = wave_reserve(n_spill, (active && spill) ? 1u : 0u);
    if (!active) return;
    if (spill) { spill_ids[sp] = pair; return; }
    const bool fits = at + want <= cx.cig_pool_cap;
    if (!fits) atomicOr(pool_over, 1u);
    const uint32_t off[2] = {at, at + (uint32_t)n_cig[0]};
    finish_records(cx, st, rd, recs + (int64_t)pair * nr, fits ? cx.cig_pool : nullptr, off, n_cig, detail2);
    PairOut o;
    o.flags = h.flags; o.est = h.est; o.est_lo = h.est_lo; o.est_hi = h.est_hi;
    o.pair_dist = h.pair_dist; o.pair_ok = (int16_t)h.pair_ok; o.mapped = (int16_t)h.mapped;
    pout[pair] = o;
}

#endif // __HIPCC__

} // namespace mcx
#endif
