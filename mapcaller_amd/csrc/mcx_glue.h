// mapcaller_amd/csrc/mcx_glue.h — the per-pair logic between the FM-index walk and the DP
// kernels, and after them: seed sorting, clustering, mate pairing, mate rescue, fragment
// construction (which decides what DP jobs exist), post-DP gates, scoring, flags, MAPQ, CIGAR.
//
// In the reference this is host C++ over std::vector/std::string (ReadMapping.cpp,
// ReadAlignment.cpp, AlignmentRescue.cpp, KmerAnalysis.cpp, SamReport.cpp, tools.cpp).  Here
// it is allocation-free code over the fixed-capacity pair-state record (mcx_types.h), one pair
// per lane, so that the whole path from reads to alignment records stays on the GPU.  Gapped
// fragments are never materialised as strings: a fragment is (rPos,rLen,gPos,gLen,kind) plus,
// for DP results, a column string of 'M'/'I'/'D' in the pair's ops pool; bases are fetched from
// the encoded read and the 2-bit genome when a column has to be compared.
//
// Every function names the reference lines it replaces; results must be bit-identical.
#ifndef MCX_GLUE_H
#define MCX_GLUE_H
#include "mcx_fm.h"

namespace mcx {

struct Ctx {
    IndexView ix;
    Params pm;
    Caps caps;
    Layout lay;
    uint8_t *detail;          // per-read alignment detail (null unless the profile is being kept)
    DetailLayout dlay;
    uint8_t *state;           // pair-state records
    const uint8_t *mapq_tab;  // [(rlen_max+1) * 6]: EvaluateMAPQ for (score, score-sub in 1..5), host-computed
    int32_t mapq_rows;
    // CIGAR operations of a batch go to one pool (BAM-style words); a read's record holds the offset of its first one
    uint32_t *cig_pool;
    uint32_t *cig_pool_n;     // words taken so far (device: reserved per wave with one atomic)
    uint32_t cig_pool_cap;
    // the batch's reads as 2-bit words (k_pack_reads; null on the host) and, per read, bit 31 = it holds an N (k_seed)
    const uint32_t *packed;
    int32_t wpad;
    const uint32_t *read_ext;
    int32_t dp_summary;       // the DP kernels leave a DpSummary in front of every problem's columns (stage_build reserves the room)
    const Hit *seed_pool;     // the seeds of the candidates mate rescue added window by window (Cand::in_pool); null on the host
};

constexpr int kCigStage = 4; // CIGAR operations per read that the finish stage keeps at hand between counting and writing them

static inline MCX_HD int64_t hit_pd(const Hit &h) { return h.gPos - h.rPos; }

// ------------------------------------------------------------------------------------------
// seeds -> candidates
// ------------------------------------------------------------------------------------------
// tail of IdentifySimplePairs (ReadMapping.cpp:141-152): keep PosDiff > 0, sort by (PosDiff, rPos)
static inline MCX_HD int prep_seeds(Hit *h, int n)
{
    // (an element is stored only when it moves: most reads have two or three seeds, already in order)
    int m = 0;
    bool in_order = true; // the kept seeds came in (PosDiff, rPos) order: nothing to sort (a read's two or three seeds of one locus)
    int64_t ppd = 0;
    int prp = 0;
    for (int i = 0; i < n; i++) {
        const Hit x = h[i];
        const int64_t pd = hit_pd(x);
        if (pd > 0) {
            if (m > 0 && (ppd > pd || (ppd == pd && prp > x.rPos))) in_order = false;
            if (m != i) h[m] = x;
            m++; ppd = pd; prp = x.rPos;
        }
    }
    if (in_order) return m;
    // insertion sort; long lists (reads from repeats: hundreds of seeds in suffix-array order) first in strides, so that
    // no element travels far one step at a time.  Equal keys are equal seeds: any order of them is the reference's.
    for (int gap = m > 24 ? (m > 400 ? 109 : 23) : 1; gap >= 1; gap = gap > 23 ? 23 : (gap > 5 ? 5 : (gap > 1 ? 1 : 0))) {
        for (int i = gap; i < m; i++) {
            const Hit key = h[i];
            const int64_t kpd = hit_pd(key);
            int j = i - gap;
            while (j >= 0) {
                const Hit y = h[j];
                const int64_t pd = hit_pd(y);
                if (pd > kpd || (pd == kpd && y.rPos > key.rPos)) { h[j + gap] = y; j -= gap; } else break;
            }
            if (j + gap != i) h[j + gap] = key;
        }
    }
    return m;
}

static inline MCX_HD int64_t boundary_of(const IndexView &ix, int64_t gPos) // GetAlignmentBoundary, tools.cpp:112-117
{
    int s = end_slot(ix, gPos);
    return s < 0 ? -1 : ix.end_pos[s];
}

static inline MCX_HD void cand_init(Cand &dst, int score, int first, int count, int64_t pd0)
{
    Cand c;
    c.score = score; c.mate = -1; c.first = first; c.count = count; c.pd0 = pd0;
    c.frag_off = 0; c.n_frags = 0; c.flag = 0; c.fwd = 1; c.in_pool = 0; c.pad = 0; c.pool_off = 0;
    dst = c; // two 16-byte stores
}

// SimplePairClustering (ReadMapping.cpp:194-226) with IdentifyClosestFragmentPairs (:160-192).
// The terminal fragment pair the reference appends is index n here (gPos = PosDiff = 2G).
// (emit(k, score, first, count, pd0): what becomes of cluster number k)
template <class Emit>
static inline MCX_HD int cluster_seeds_to(const IndexView &ix, const Params &pm, int rlen, const Hit *h, int n, Emit emit)
{
    if (n == 0) return 0;
    const Hit h0 = h[0];
    int nc = 0, head = 0, score = h0.len, thr = rlen >> 2;
    int64_t g_end = boundary_of(ix, h0.gPos);
    int64_t pd_prev = hit_pd(h0); // of seed j - 1: every seed is fetched once
    for (int j = 1; j <= n; j++) {
        Hit hj; hj.gPos = ix.G2; hj.rPos = 0; hj.len = 0;
        if (j < n) hj = h[j];
        const int64_t gj = hj.gPos, pdj = j < n ? hit_pd(hj) : ix.G2;
        int64_t d = pdj - pd_prev;
        pd_prev = pdj;
        if (d < 0) d = -d;
        if (gj > g_end || d > pm.max_pos_diff) {
            if (score > thr) {
                if (thr < (score >> 1)) thr = score >> 1;
                int b = head, e = j, s = score;
                if (score >= rlen) { // tandem repeats: best run of equal PosDiff
                    int run_b = head, rs = h[head].len;
                    s = 0; b = e = head;
                    int k;
                    for (k = head + 1; k < j; k++) {
                        if (hit_pd(h[k]) != hit_pd(h[run_b])) {
                            if (rs > s) { s = rs; b = run_b; e = k; }
                            run_b = k; rs = h[k].len;
                        } else rs += h[k].len;
                    }
                    if (rs > s) { s = rs; b = run_b; e = k; }
                }
                emit(nc, s, b, e - b, hit_pd(h[b]));
                nc++;
            }
            head = j;
            if (j < n) { g_end = boundary_of(ix, gj); score = hj.len; }
        } else score += hj.len;
    }
    return nc;
}

static inline MCX_HD int cluster_seeds(const IndexView &ix, const Params &pm, int rlen, const Hit *h, int n,
                                       Cand *out, int cap)
{
    return cluster_seeds_to(ix, pm, rlen, h, n, [&](int k, int score, int first, int count, int64_t pd0) { if (k < cap) cand_init(out[k], score, first, count, pd0); });
}

static inline MCX_HD void keep_top_scores(Cand *c, int n) // RemoveRedundantAlnCan, ReadMapping.cpp:228-242
{
    if (n <= 1) return;
    int best = 0;
    for (int i = 0; i < n; i++) if (c[i].score > best) best = c[i].score;
    for (int i = 0; i < n; i++) if (c[i].score < best) c[i].score = 0;
}

// CheckPairedAlignmentDistance (ReadMapping.cpp:244-303).  Also reports the interval of
// EstiDistance values [lo, hi] for which every `myDist < EstiDistance` test, hence the whole
// outcome, is unchanged — that is what lets a batch run ahead of the reference's per-chunk
// avgDist feedback (ReadMapping.cpp:539) and still be replayed exactly.
static inline MCX_HD int pair_by_distance(int64_t est, Cand *c1, int n1, Cand *c2, int n2, int &lo, int &hi)
{
    int64_t max_lt = -1, min_ge = 0x7fffffff;
    if (n1 * n2 > 100) { keep_top_scores(c1, n1); keep_top_scores(c2, n2); }
    // the partner a candidate of read 1 picks among read 2's: found twice (once for the best pair sum, once to
    // mark the pairs that reach it) rather than parked in the candidate — fetches of lines already at hand
    // are cheaper here than stores.  Both lists come in ascending PosDiff (cluster_seeds_to emits its clusters in the order of
    // the sorted seeds), so read 2's candidates below a's PosDiff are passed once for all of read 1's (j0 only moves on), and
    // behind the first live one at or beyond the estimate nothing can change the outcome: the same picks, the same [lo, hi]
    // as the scan of all n1 x n2 pairs, in n1 + n2 + (pairs within the estimate) steps.
    int j0 = 0;
    auto partner = [&](const Cand &a, int &ps) {
        int pick = -1;
        ps = 0;
        while (j0 < n2 && c2[j0].pd0 < a.pd0) j0++;
        for (int j = j0; j < n2; j++) {
            const int sj = c2[j].score;
            if (sj == 0) continue;
            const int64_t d = c2[j].pd0 - a.pd0;
            if (d < est) {
                if (d > max_lt) max_lt = d;
                if (sj > ps) { pick = j; ps = sj; }
            } else { if (d < min_ge) min_ge = d; break; }
        }
        return pick;
    };
    int64_t top = 0;
    for (int i = 0; i < n1; i++) {
        const Cand a = c1[i];
        if (a.score == 0) continue;
        int ps;
        if (partner(a, ps) >= 0) { const int64_t s = (int64_t)a.score + ps; if (s > top) top = s; }
    }
    int paired = 0;
    if (top > 0) {
        j0 = 0;
        for (int i = 0; i < n1; i++) {
            const Cand a = c1[i];
            if (a.score == 0) continue;
            int ps;
            const int pick = partner(a, ps);
            if (pick >= 0 && (int64_t)a.score + ps == top) { paired++; c1[i].mate = pick; c2[pick].mate = i; }
        }
    }
    lo = (int)(max_lt + 1);
    hi = (int)min_ge;
    return paired;
}

static inline MCX_HD void mask_unpaired(Cand *c1, int n1, Cand *c2, int n2) // MaskUnPairedAlnCan, ReadMapping.cpp:305-322
{
    int top = 0;
    for (int i = 0; i < n1; i++)
        if (c1[i].mate != -1 && top < c1[i].score + c2[c1[i].mate].score) top = c1[i].score + c2[c1[i].mate].score;
    for (int i = 0; i < n1; i++)
        if (c1[i].mate == -1 || c1[i].score + c2[c1[i].mate].score < top) c1[i].score = 0;
    for (int j = 0; j < n2; j++)
        if (c2[j].mate == -1 || c2[j].score + c1[c2[j].mate].score < top) c2[j].score = 0;
}

// ------------------------------------------------------------------------------------------
// mate rescue (AlignmentRescue.cpp:28-111, KmerAnalysis.cpp:57-163)
// ------------------------------------------------------------------------------------------
// The reference sorts 8-mer lists and joins them; the join result, ordered by (PosDiff, rPos),
// is exactly a scan of the (read position x window position) match matrix along diagonals, so
// that is what is done here, on 8-mer ids laid out by position.
#define MCX_NOKMER 0xFFFFFFFFu

// the character the reference's k-mer builder sees at position i of a read
static inline MCX_HD uint8_t read_char(const ReadRef &r, int i)
{
    if (!r.flipped) return r.ascii[i];
    switch (r.ascii[r.rlen - 1 - i]) { // GetComplementaryBase, tools.cpp:3-18
    case 'A': case 'a': return 'T';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    case 'T': case 't': return 'A';
    default: return 'N';
    }
}

template <class GetCh>
static inline MCX_HD void kmer_fill(GetCh ch, int len, uint32_t *out) // CreateKmerVecFromReadSeq, KmerAnalysis.cpp:57-103
{
    for (int i = 0; i < len; i++) out[i] = MCX_NOKMER;
    uint32_t tail = 0, count = 0, head, wid;
    while (count < (uint32_t)kKmerSize && tail < (uint32_t)len) { if (ch(tail++) != 'N') count++; else count = 0; }
    if (count != (uint32_t)kKmerSize) return;
    head = tail - kKmerSize;
    wid = 0;
    for (uint32_t i = head; i < head + kKmerSize; i++) wid = (wid << 2) + (uint32_t)nt4_code(ch(i));
    out[head] = wid;
    for (head += 1; tail < (uint32_t)len; head++, tail++) {
        uint8_t c = ch(tail);
        if (c != 'N') {
            wid = ((wid & 0x3FFFu) << 2) + (uint32_t)nt4_code(c);
            out[head] = wid;
        } else {
            count = 0; tail++;
            while (count < (uint32_t)kKmerSize && tail < (uint32_t)len) { if (ch(tail++) != 'N') count++; else count = 0; }
            if (count != (uint32_t)kKmerSize) break;
            head = tail - kKmerSize;
            wid = 0;
            for (uint32_t i = head; i < head + kKmerSize; i++) wid = (wid << 2) + (uint32_t)nt4_code(ch(i));
            out[head] = wid;
        }
    }
}

struct RescueOut { int score; int n_seeds; int d; };

// one diagonal d of the (read position x window position) 8-mer match matrix: runs of
// consecutive matches of length >= 3 are seeds of 8 + run - 1 bases
// (GenerateSimplePairsFromCommonKmers with thr = 10, KmerAnalysis.cpp:133-163).  Returns the
// summed seed length; with hits != null the seeds are appended (emitted counts them).
static inline MCX_HD int diag_scan(const uint32_t *kq, int qlen, const uint32_t *kg, int slen, int d, int64_t base,
                                   Hit *hits, int n_hits, int cap, int &emitted, bool &overflow)
{
    const int r0 = d < 0 ? -d : 0, r1 = qlen - 1 < slen - 1 - d ? qlen - 1 : slen - 1 - d;
    int total = 0, run = 0, run_start = 0;
    for (int r = r0; r <= r1 + 1; r++) {
        const bool m = r <= r1 && kq[r] != MCX_NOKMER && kq[r] == kg[r + d];
        if (m) { if (run == 0) run_start = r; run++; }
        else if (run > 0) {
            const int l = kKmerSize + run - 1;
            if (l >= 10) {
                total += l;
                if (hits) {
                    if (n_hits + emitted < cap) {
                        Hit h; h.rPos = run_start; h.gPos = (int64_t)(run_start + d) + base; h.len = l;
                        hits[n_hits + emitted] = h;
                    } else overflow = true;
                    emitted++;
                }
            }
            run = 0;
        }
    }
    return total;
}

// The same total for one diagonal without a chain of dependent steps: the diagonal's matches as a bit string, 61 new
// bits at a time behind 3 bits of history.  A run of L >= 3 matches contributes L + 7; it holds L - 2 "triples" (three
// matches in a row), so the sum is (number of triples) + 9 x (number of runs that hold one).  A triple is counted in the
// word its last match falls into; a run starts where a triple has no triple right before it.
static inline MCX_HD int diag_total(const uint32_t *kq, int qlen, const uint32_t *kg, int slen, int d)
{
    const int r0 = d < 0 ? -d : 0, r1 = qlen - 1 < slen - 1 - d ? qlen - 1 : slen - 1 - d;
    int total = 0;
    uint64_t hist = 0; // the last three matches before the word, bit 2 the most recent
    for (int base = r0; base <= r1; base += 61) {
        uint64_t e = hist;
        const int n = r1 - base + 1 < 61 ? r1 - base + 1 : 61;
        for (int i = 0; i < n; i++) {
            const uint32_t q = kq[base + i];
            e |= (uint64_t)(q != MCX_NOKMER && q == kg[base + i + d]) << (3 + i);
        }
        const uint64_t t = e & (e >> 1) & (e >> 2);   // bit i: matches at i, i+1, i+2
        const uint64_t own = t & ~(uint64_t)1;        // (the triple at bit 0 ended in the word before)
#if defined(__HIP_DEVICE_COMPILE__)
        total += __popcll(own) + 9 * __popcll(own & ~(t << 1));
#else
        total += __builtin_popcountll(own) + 9 * __builtin_popcountll(own & ~(t << 1));
#endif
        hist = (e >> 61) & 7u;
    }
    return total;
}

// Serial evaluation of one rescue window (host emulation, and the reference semantics the
// wave-cooperative version below must reproduce): IdentifyCommonKmers +
// GenerateSimplePairsFromCommonKmers + IdentifyBestAlnCan.
struct RescueSerial {
    uint32_t *kq, *kg;
    MCX_HD bool leader() const { return true; }
    MCX_HD void sync() const {}
    MCX_HD void fill_query(const ReadRef &rq) const
    {
        kmer_fill([&](uint32_t i) { return read_char(rq, (int)i); }, rq.rlen, kq);
    }
    MCX_HD RescueOut window(const IndexView &ix, int64_t left, int slen, int qlen, Hit *hits, int n_hits, int cap, bool &overflow) const
    {
        kmer_fill([&](uint32_t i) { return (uint8_t)"ACGT"[ref_code(ix, left + i)]; }, slen, kg);
        RescueOut best; best.score = 0; best.n_seeds = 0; best.d = 0;
        int dummy = 0;
        for (int d = -(qlen - 1); d <= slen - 1; d++) {
            const int total = diag_scan(kq, qlen, kg, slen, d, left, nullptr, 0, 0, dummy, overflow);
            if (total > best.score) { best.score = total; best.d = d; }
        }
        if (best.score > 0) diag_scan(kq, qlen, kg, slen, best.d, left, hits, n_hits, cap, best.n_seeds, overflow);
        return best;
    }
};

#if defined(__HIPCC__)
// The same evaluation by one workgroup, on bit planes.  Two 8-mers are equal when their eight bases are, so the 8-mer match
// matrix of a diagonal is "eight base matches in a row": read and window are kept in LDS as two bit planes each (the low
// and the high bit of every base, 32 bases to a word; the genome has no N, a read's N are a third plane), a diagonal's
// base matches are ~((QL ^ WL') | (QH ^ WH')) with the window planes funnel-shifted to the diagonal — 32 rows per handful of
// instructions instead of one id comparison per cell — and runs of eight collapse with three shift-and-AND steps.  The
// diagonal's total then comes from the 8-mer match bits as in diag_total: a run of L >= 3 matches counts L + 7 =
// (its L - 2 triples) + 9.  Words are walked from the read's end to its start so that what a word needs from the next one
// (the carries of the collapses, the triple bits) is at hand.
constexpr int kRescueQWords = 1024 / 32 + 2;                 // plane words of the longest read (+ the funnel's look-ahead)
constexpr int rescue_wwords(int kg) { return (kg + 2 * 1056) / 32 + 4; } // window plane: pad | window | the read's overhang

static __device__ __forceinline__ uint32_t even_bits16(uint32_t t) // bits 0,2,..30 of t, packed into 16
{
    t &= 0x55555555u; t = (t | (t >> 1)) & 0x33333333u; t = (t | (t >> 2)) & 0x0f0f0f0fu; t = (t | (t >> 4)) & 0x00ff00ffu;
    return (t | (t >> 8)) & 0xffffu;
}

struct RescueWave { // one workgroup (any number of wavefronts) evaluates one pair
    uint32_t *q;    // LDS: QL | QH | QV, kRescueQWords each (QV: an 8-mer of the read starts here — inside the read, no N)
    uint32_t *w;    // LDS: WL | WH, wstride words each; window position p is bit p + pad
    uint32_t *ew;   // LDS: the 8-mer match words of the best diagonal (kRescueQWords)
    int wstride;
    int *red;       // LDS scratch: 2 ints per wavefront + 4, then the flag "the read has N"
    uint32_t *kq, *kg; // HBM scratch of the workgroup for window_ids: 1024 ids of the read; ids of the window + its 2-bit bytes
    __device__ bool leader() const { return threadIdx.x == 0; }
    __device__ void sync() const { __threadfence_block(); __syncthreads(); }
    __device__ void fill_query(const ReadRef &rq) const
    {
        const int tid = threadIdx.x, nt = blockDim.x;
        uint32_t *ql = q, *qh = q + kRescueQWords, *qn = q + 2 * kRescueQWords;
        __syncthreads();
        for (int i = tid; i < 3 * kRescueQWords; i += nt) q[i] = 0u;
        if (tid == 0) red[0] = 0; // window()'s best-diagonal key
        __syncthreads();
        for (int i = tid; i < rq.rlen; i += nt) {
            const int c = nt4_code(read_char(rq, i));
            const uint32_t bit = 1u << (i & 31);
            if (c > 3) atomicOr(&qn[i >> 5], bit);
            else { if (c & 1) atomicOr(&ql[i >> 5], bit); if (c & 2) atomicOr(&qh[i >> 5], bit); }
        }
        __syncthreads();
        int any_n = 0;
        for (int k = tid; k < kRescueQWords; k += nt) any_n |= qn[k] != 0u;
        any_n = __syncthreads_or(any_n);
        if (tid == 0) red[2 * (nt >> 6) + 4] = any_n;
        if (any_n) { // (window_ids will be used)
            if (leader()) kmer_fill([&](uint32_t i) { return read_char(rq, (int)i); }, rq.rlen, kq);
            __threadfence_block();
            __syncthreads();
            return;
        }
        // a base counts when it is inside the read and not N; an 8-mer starts where eight of them follow one another
        uint32_t v8 = 0;
        if (tid < kRescueQWords) {
            auto ok = [&](int k) -> uint32_t {
                if (k >= kRescueQWords || 32 * k >= rq.rlen) return 0u;
                const uint32_t in = rq.rlen - 32 * k >= 32 ? ~0u : ((1u << (rq.rlen - 32 * k)) - 1u);
                return ~qn[k] & in;
            };
            const uint32_t v = ok(tid), vn = ok(tid + 1);
            v8 = v;
#pragma unroll
            for (int j = 1; j < kKmerSize; j++) v8 &= __funnelshift_r(v, vn, j);
        }
        __syncthreads();
        if (tid < kRescueQWords) qn[tid] = v8;
        __syncthreads();
    }
    // the 8-mer match words of diagonal d, from the read's last word down; total: the diagonal's summed seed length
    // (store: the words are kept in ew for the seeds)
    __device__ int diagonal(int d, int qlen, int slen, int pad, bool store) const
    {
        const uint32_t *ql = q, *qh = q + kRescueQWords, *qv = q + 2 * kRescueQWords, *wl = w, *wh = w + wstride;
        const int lo = d < 0 ? -d : 0, hi = qlen - kKmerSize < slen - kKmerSize - d ? qlen - kKmerSize : slen - kKmerSize - d; // rows whose 8-mers exist on both sides
        if (hi - lo < 2) return 0;
        const int ktop = (hi + kKmerSize - 1) >> 5, kbot = lo >> 5;
        const int s0 = d + pad, sh = s0 & 31;
        int i = ktop + (s0 >> 5);
        uint32_t wl_hi = wl[i + 1], wh_hi = wh[i + 1], m_hi = 0, a_hi = 0, b_hi = 0, e_hi = 0, t_hi = 0;
        int total = 0;
        for (int k = ktop; k >= kbot; k--, i--) {
            const uint32_t wl_lo = wl[i], wh_lo = wh[i];
            const uint32_t gl = __funnelshift_r(wl_lo, wl_hi, sh), gh = __funnelshift_r(wh_lo, wh_hi, sh);
            wl_hi = wl_lo; wh_hi = wh_lo;
            const uint32_t m = ~((ql[k] ^ gl) | (qh[k] ^ gh));          // bases equal, rows 32k..32k+31
            const uint32_t a = m & __funnelshift_r(m, m_hi, 1);
            const uint32_t b = a & __funnelshift_r(a, a_hi, 2);
            const uint32_t c = b & __funnelshift_r(b, b_hi, 4);          // eight in a row from this row on
            m_hi = m; a_hi = a; b_hi = b;
            const int x0 = lo - 32 * k, x1 = hi - 32 * k;
            const uint32_t rows = (x1 < 0 || x0 > 31) ? 0u : ((~0u << (x0 > 0 ? x0 : 0)) & (~0u >> (31 - (x1 < 31 ? x1 : 31))));
            const uint32_t e = c & qv[k] & rows;
            if (store) ew[k] = e;
            uint32_t t = 0;
            if (e | e_hi) {
                t = e & __funnelshift_r(e, e_hi, 1) & __funnelshift_r(e, e_hi, 2); // a triple starts here
                total += __popc(t) + 9 * __popc(t & ~__funnelshift_r(t, t_hi, 1));   // + 9 per run (counted where its last triple starts)
            }
            t_hi = t; e_hi = e;
        }
        return total;
    }
    // the same evaluation on 8-mer ids, for a read with N: CreateKmerVecFromReadSeq's rolling id falls out of step with the
    // positions after an N (KmerAnalysis.cpp:81-95 skips a character when it resumes), so its ids are not the read's plain
    // 8-mers there and the planes cannot stand in for them.  kq / kg: the workgroup's scratch in HBM (such reads are rare).
    __device__ RescueOut window_ids(const IndexView &ix, int64_t left, int slen, int qlen, Hit *hits, int n_hits, int cap, bool &overflow) const
    {
        const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, n_waves = nt >> 6;
        __syncthreads();
        // The window's bases come from the 2-bit genome: its bytes are copied into LDS once (behind the ids: gb) and the 8-mer
        // ids are cut out of them, instead of eight byte fetches from HBM per position.  A window on the reverse strand is the
        // mirrored forward stretch, complemented.
        uint8_t *gb = (uint8_t *)(kg + slen + 8);
        if (left < ix.G && left + slen > ix.G) { // the window runs from the end of the forward strand into the reverse strand
            for (int p = tid; p < slen; p += nt) { // (same chromosome on both sides of G: AlignmentRescue lets it pass): base by base
                uint32_t wid = MCX_NOKMER;
                if (p + kKmerSize <= slen) { wid = 0; for (int k = 0; k < kKmerSize; k++) wid = (wid << 2) | (uint32_t)ref_code(ix, left + p + k); }
                kg[p] = wid;
            }
        } else {
        const bool rev = left >= ix.G;
        const int64_t f0 = rev ? ix.G2 - (left + slen) : left;
        const int64_t b0 = f0 >> 2;
        const int n_bytes = (int)(((f0 + slen - 1) >> 2) - b0) + 1;
        for (int i = tid; i < n_bytes; i += nt) gb[i] = ix.pac[b0 + i];
        __syncthreads();
        auto base = [&](int p) -> uint32_t { // code of window position p
            const int64_t f = rev ? f0 + (slen - 1 - p) : f0 + p;
            const uint32_t c = (gb[(f >> 2) - b0] >> ((~f & 3) << 1)) & 3u;
            return rev ? 3u - c : c;
        };
        for (int p = tid; p < slen; p += nt) {
            uint32_t wid = MCX_NOKMER;
            if (p + kKmerSize <= slen) {
                wid = 0;
                for (int k = 0; k < kKmerSize; k++) wid = (wid << 2) | base(p + k);
            }
            kg[p] = wid;
        }
        }
        __syncthreads();
        int best_total = 0, best_d = 0x7fffffff;
        // Four neighbouring diagonals per thread, walked together along the read: the window ids they need at read
        // position r are four consecutive words of kg, three of them already at hand from r - 1 — one LDS fetch of
        // kg and one (broadcast) of kq per four cells.  The matches of a diagonal collect as a bit string, 29 new bits
        // behind 3 of history (diag_total's bookkeeping in 32-bit words).
        constexpr int D = 4;
        const int n_groups = (qlen + slen - 1 + D - 1) / D;
        for (int g = tid; g < n_groups; g += nt) {
            const int dbase = -(qlen - 1) + g * D;
            uint32_t e[D], win[D];
            int tot[D];
#pragma unroll
            for (int j = 0; j < D; j++) { e[j] = 0u; tot[j] = 0; }
            auto kgv = [&](int p) -> uint32_t { return (p >= 0 && p < slen) ? kg[p] : MCX_NOKMER; };
            const int rlo = -(dbase + D - 1) > 0 ? -(dbase + D - 1) : 0, rhi = qlen - 1 < slen - 1 - dbase ? qlen - 1 : slen - 1 - dbase;
#pragma unroll
            for (int j = 0; j + 1 < D; j++) win[j] = kgv(rlo + dbase + j);
            for (int r = rlo; r <= rhi;) {
                for (int i = 0; i < 29 && r <= rhi; i++, r++) {
                    uint32_t q = kq[r];
                    if (q == MCX_NOKMER) q = 0xFFFFFFFEu; // (matches nothing: ids are 16 bits, an absent window id is MCX_NOKMER)
                    win[D - 1] = kgv(r + dbase + D - 1);
#pragma unroll
                    for (int j = 0; j < D; j++) e[j] |= (uint32_t)(q == win[j]) << (3 + i);
#pragma unroll
                    for (int j = 0; j + 1 < D; j++) win[j] = win[j + 1];
                }
#pragma unroll
                for (int j = 0; j < D; j++) {
                    const uint32_t t = e[j] & (e[j] >> 1) & (e[j] >> 2), own = t & ~1u;
                    tot[j] += __popc(own) + 9 * __popc(own & ~(t << 1));
                    e[j] = (e[j] >> 29) & 7u;
                }
            }
#pragma unroll
            for (int j = 0; j < D; j++)
                if (dbase + j <= slen - 1 && tot[j] > best_total) { best_total = tot[j]; best_d = dbase + j; } // (ascending: the first of equal totals stays)
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int ot = __shfl_xor(best_total, o, 64), od = __shfl_xor(best_d, o, 64);
            if (ot > best_total || (ot == best_total && od < best_d)) { best_total = ot; best_d = od; }
        }
        if (lane == 0) { red[2 * wave] = best_total; red[2 * wave + 1] = best_d; }
        __syncthreads();
        int *out = red + 2 * n_waves; // {score, d, n_seeds, overflow}
        if (tid == 0) {
            for (int w = 1; w < n_waves; w++) {
                const int ot = red[2 * w], od = red[2 * w + 1];
                if (ot > best_total || (ot == best_total && od < best_d)) { best_total = ot; best_d = od; }
            }
            int n_seeds = 0;
            bool o2 = false;
            if (best_total > 0) diag_scan(kq, qlen, kg, slen, best_d, left, hits, n_hits, cap, n_seeds, o2);
            out[0] = best_total; out[1] = best_total > 0 ? best_d : 0; out[2] = n_seeds; out[3] = o2 ? 1 : 0;
        }
        __syncthreads();
        RescueOut best; best.score = out[0]; best.d = out[1]; best.n_seeds = out[2];
        if (out[3]) overflow = true;
        return best;
    }

    __device__ RescueOut window(const IndexView &ix, int64_t left, int slen, int qlen, Hit *hits, int n_hits, int cap, bool &overflow) const
    {
        const int tid = threadIdx.x, nt = blockDim.x, n_waves = nt >> 6;
        if (red[2 * n_waves + 4]) return window_ids(ix, left, slen, qlen, hits, n_hits, cap, overflow);
        uint32_t *wl = w, *wh = w + wstride;
        const int pad = ((qlen + 31) & ~31) + 32;                 // diagonals reach qlen - 1 positions before the window
        const int n_words = (pad + slen + qlen) / 32 + 3;
        __syncthreads();
        // The window's planes, a word (32 positions) per thread: two 16-base stretches of the 2-bit genome (ref_codes16: either
        // strand, and the few windows that run from one into the other), their low and high bits pulled apart.  Words before
        // and behind the window are clear.
        for (int j = tid; j < n_words; j += nt) {
            uint32_t lo = 0, hi = 0;
            const int p0 = 32 * j - pad; // window position of the word's bit 0
            if (p0 >= 0 && p0 < slen) {
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int p = p0 + 16 * half;
                    if (p >= slen) break;
                    const uint32_t x = ref_codes16(ix, left + p); // base s at bits 31-2s (high), 30-2s (low)
                    uint32_t l16 = __brev(even_bits16(x)) >> 16, h16 = __brev(even_bits16(x >> 1)) >> 16; // base s at bit s
                    if (slen - p < 16) { const uint32_t keep = (1u << (slen - p)) - 1u; l16 &= keep; h16 &= keep; }
                    lo |= l16 << (16 * half); hi |= h16 << (16 * half);
                }
            }
            wl[j] = lo; wh[j] = hi;
        }
        __syncthreads();
        // a diagonal needs three 8-mers in a row on both sides to score at all: d in [-(qlen - 10), slen - 10].
        // The best one (largest total, then smallest diagonal) is found with one LDS atomic per thread, on a key made of both.
        const int d_lo = -(qlen - kKmerSize - 2), n_diag = slen - kKmerSize - 2 - d_lo + 1;
        uint32_t my_key = 0;
        for (int g = tid; g < n_diag; g += nt) {
            const int total = diagonal(d_lo + g, qlen, slen, pad, false);
            const uint32_t key = ((uint32_t)total << 13) | (uint32_t)(8191 - g); // (g < 8192: a window is at most 4096 long, a read 1000)
            if (total > 0 && key > my_key) my_key = key;
        }
        uint32_t *best_key = (uint32_t *)red; // red[0]: the key (cleared again below); red[1], red[2]: seeds written, overflow
        if (my_key) atomicMax(best_key, my_key);
        __syncthreads();
        const uint32_t key = *best_key;
        RescueOut best; best.score = (int)(key >> 13); best.d = 0; best.n_seeds = 0;
        if (key == 0) return best; // (the key is still clear for the next window, whose first barrier comes before anybody touches it)
        best.d = d_lo + (8191 - (int)(key & 8191u));
        if (tid == 0) {
            const int best_d = best.d;
            int n_seeds = 0;
            bool o2 = false;
            // the seeds of the best diagonal, in read order (diag_scan on the match bits)
            for (int k = 0; k < kRescueQWords; k++) ew[k] = 0u;
            diagonal(best_d, qlen, slen, pad, true);
            int run = 0, run_start = 0;
            for (int r = 0; r <= qlen; r++) {
                if (run == 0 && (r & 31) == 0 && r + 32 <= qlen && ew[r >> 5] == 0u) { r += 31; continue; } // (a word without matches)
                const bool m = r < qlen && ((ew[r >> 5] >> (r & 31)) & 1u);
                if (m) { if (run == 0) run_start = r; run++; }
                else if (run > 0) {
                    const int l = kKmerSize + run - 1;
                    if (l >= 10) {
                        if (n_hits + n_seeds < cap) {
                            Hit h; h.rPos = run_start; h.gPos = (int64_t)(run_start + best_d) + left; h.len = l;
                            hits[n_hits + n_seeds] = h;
                        } else o2 = true;
                        n_seeds++;
                    }
                    run = 0;
                }
            }
            red[1] = n_seeds; red[2] = o2 ? 1 : 0;
        }
        __syncthreads();
        best.n_seeds = red[1];
        if (red[2]) overflow = true;
        if (tid == 0) *best_key = 0u; // (every thread has read it; clear for the next window)
        return best;
    }
};
#endif

// AlignmentRescue (AlignmentRescue.cpp:28-111).  Every lane of the evaluating wave (or the one
// host thread) follows the same control flow over the pair state; only the leader writes it.
// Windows that leave [0,2G) make the reference read outside RefSequence (undefined behaviour,
// it crashes near the genome start); they are skipped.
template <class Eval>
static inline MCX_HD int rescue_mate(const Ctx &cx, PairState &st, const ReadRef &r1, const ReadRef &r2, uint32_t est,
                                     const Eval &ev)
{
    const IndexView &ix = cx.ix;
    PairHdr &h = *st.hdr;
    int n1 = h.n_cands[0], n2 = h.n_cands[1];
    int nh[2] = {h.n_hits[0], h.n_hits[1]};
    uint32_t add_flags = 0;
    Cand *c1 = st.cands[0], *c2 = st.cands[1];
    int s1 = 0, s2 = 0, paired = 0;
    for (int i = 0; i < n1; i++) if (c1[i].score > s1) s1 = c1[i].score;
    for (int i = 0; i < n2; i++) if (c2[i].score > s2) s2 = c2[i].score;
    int mode;
    if (s1 < (r1.rlen >> 2) && s2 < (r2.rlen >> 2)) return 0;
    else if (s1 - s2 > (r2.rlen >> 2)) mode = 1;
    else if (s2 - s1 > (r1.rlen >> 2)) mode = 2;
    else mode = 3;
    add_flags |= kRescueUsedEst;
    for (int side = 0; side < 2; side++) {
        if (side == 0 && !(mode == 1 || mode == 3)) continue;
        if (side == 1 && !(mode == 2 || mode == 3)) continue;
        // side 0: place read2 next to read1's candidates; side 1: the other way round
        const ReadRef &rq = side == 0 ? r2 : r1;
        Cand *ca = side == 0 ? c1 : c2, *cb = side == 0 ? c2 : c1;
        int &na = side == 0 ? n1 : n2, &nb = side == 0 ? n2 : n1;
        const int sa = side == 0 ? s1 : s2, sb = side == 0 ? s2 : s1;
        Hit *hb = st.hits[side == 0 ? 1 : 0];
        int &nhb = nh[side == 0 ? 1 : 0];
        const int thr = sa >> 1, lim = na;
        bool filled = false;
        for (int ci = 0; ci < lim; ci++) {
            const Cand c = ca[ci];
            if (c.score < thr || c.mate != -1) continue;
            const int64_t left = side == 0 ? c.pd0 : c.pd0 - (int64_t)est;
            int64_t right = side == 0 ? c.pd0 + est + rq.rlen : c.pd0 + rq.rlen;
            if (right > ix.G2) right = ix.G2;
            if (left < 0 || right >= ix.G2) continue;
            const int e1 = end_slot(ix, left), e2 = end_slot(ix, right);
            if (e1 < 0 || e2 < 0 || ix.end_chr[e1] != ix.end_chr[e2]) continue;
            const int slen = (int)(right - left);
            if (slen < rq.rlen) continue;
            if (slen > cx.caps.kmer_cap) { add_flags |= kOvKmer; continue; }
            if (!filled) { ev.fill_query(rq); filled = true; }
            bool ov = false;
            const RescueOut ro = ev.window(ix, left, slen, rq.rlen, hb, nhb, cx.caps.hit_cap, ov);
            if (ro.n_seeds == 0) continue;
            if (ro.score > sb) {
                if (ov) { add_flags |= kOvHits; continue; }
                if (nb >= cx.caps.cand_cap) { add_flags |= kOvCands; continue; }
                paired++;
                if (ev.leader()) {
                    ca[ci].mate = nb;
                    cand_init(cb[nb], ro.score, nhb, ro.n_seeds, (int64_t)ro.d + left);
                    cb[nb].mate = ci;
                }
                nb++;
                nhb += ro.n_seeds;
                ev.sync();
            }
        }
    }
    if (ev.leader()) {
        h.flags |= add_flags;
        h.n_cands[0] = n1; h.n_cands[1] = n2;
        h.n_hits[0] = nh[0]; h.n_hits[1] = nh[1];
    }
    ev.sync();
    return paired;
}

// ------------------------------------------------------------------------------------------
// stage G1a: cluster both reads and pair them (ReadMapping.cpp:445-463)
// ------------------------------------------------------------------------------------------
static inline MCX_HD bool pair_needs_rescue(const PairHdr &h) { return h.n_paired == 0; }

// (n_hits_in: the reads' hit counts when the caller has them elsewhere — the device keeps them in a per-read
//  array, so that the seeding kernel need not touch the header and this stage need not fetch it: the stage
//  writes every field of the header that later stages read before writing)
static inline MCX_HD void stage_cluster_pair(const Ctx &cx, int64_t pair, const ReadRef *rd, int est, const int *n_hits_in = nullptr)
{
    PairState st = pair_state(cx.state, cx.lay, cx.caps, pair);
    PairHdr *const g_hdr = st.hdr;
    PairHdr h;
    if (n_hits_in) {
        h.flags = 0; h.n_frags = 0; h.n_ops = 0; h.pair_dist = 0; h.n_jobs = 0; h.pair_ok = 0; h.mapped = 0; h.pad[0] = h.pad[1] = 0;
        h.n_hits[0] = (int16_t)n_hits_in[0]; h.n_hits[1] = (int16_t)n_hits_in[1]; h.n_cands[0] = h.n_cands[1] = 0;
        h.sum[0].best = h.sum[1].best = -1; h.sum[0].score = h.sum[1].score = 0; h.sum[0].sub = h.sum[1].sub = 0;
    } else h = *g_hdr; // in registers through the stage, stored back once
    int nr = cx.pm.paired ? 2 : 1;
    MCX_UNROLL // (written to unroll: a header indexed by a run-time mate number lives in scratch memory)
    for (int s = 0; s < 2; s++) {
        if (s >= nr) break;
        if (h.n_hits[s] > cx.caps.hit_seed) { h.flags |= kOvHits; h.n_hits[s] = 0; }
        h.n_hits[s] = prep_seeds(st.hits[s], h.n_hits[s]);
        int nc = cluster_seeds(cx.ix, cx.pm, rd[s].rlen, st.hits[s], h.n_hits[s], st.cands[s], cx.caps.cand_seed);
        if (nc > cx.caps.cand_seed) { h.flags |= kOvCands; nc = 0; }
        h.n_cands[s] = nc;
        h.sum[s].best = -1; h.sum[s].score = 0; h.sum[s].sub = 0;
    }
    h.est = est; h.est_lo = 0; h.est_hi = 0x7fffffff; h.n_paired = 0;
    if (cx.pm.paired && !(h.flags & kOvAny))
        h.n_paired = pair_by_distance(est, st.cands[0], h.n_cands[0], st.cands[1], h.n_cands[1], h.est_lo, h.est_hi);
    *g_hdr = h;
}

// stage R: mate rescue for pairs left unpaired (ReadMapping.cpp:463)
template <class Eval>
static inline MCX_HD void stage_rescue(const Ctx &cx, int64_t pair, const ReadRef *rd, const Eval &ev)
{
    PairState st = pair_state(cx.state, cx.lay, cx.caps, pair);
    PairHdr &h = *st.hdr;
    if (!cx.pm.paired || (h.flags & kOvAny) || h.n_paired != 0) return;
    const int np = rescue_mate(cx, st, rd[0], rd[1], (uint32_t)h.est, ev);
    if (ev.leader()) h.n_paired = np;
    ev.sync();
}

// ------------------------------------------------------------------------------------------
// stage G1b: mask candidates, build fragment lists, decide DP jobs
// (ReadMapping.cpp:469-470, ReadAlignment.cpp:306-342 up to and including ProcessNormalPair's
// decision at :184)
// ------------------------------------------------------------------------------------------
static inline MCX_HD bool frag_before(const Frag &a, const Frag &b) // CompByReadPos, ReadAlignment.cpp:23-27
{
    return a.rPos == b.rPos ? a.gPos < b.gPos : a.rPos < b.rPos;
}

// builds the fragment list of one candidate at f[0..); returns the count, or -1 when the
// alignment would span two chromosomes (CheckAlignmentValidity, tools.cpp:119-130)
static inline MCX_HD int build_frags(const IndexView &ix, int rlen, const Hit *seeds, int n, Frag *f)
{
    int total = 0;
    // The usual case in one pass: seeds that already come in read order, apart from each other on the read and
    // on the genome, need no sorting, no overlap removal and no expansion in place — the list is written once,
    // front to back, gaps included.  Anything else starts over on the general path below (same result).
    {
        bool plain = true;
        int w = 0, prev_r0 = 0, pr = 0;
        int64_t prev_g0 = 0, pg = 0;
        Frag g; g.ops_off = 0; g.ops_len = 0; g.kind = kPlain; g.meta = 0;
        for (int i = 0; i < n; i++) {
            const Hit sd = seeds[i];
            const int r0 = sd.rPos;
            if (i == 0) {
                if (r0 > 0) { g.rPos = 0; g.gPos = sd.gPos - r0; g.rLen = g.gLen = r0; f[w++] = g; }
            } else {
                const int rg = r0 - pr;
                const int64_t gg = sd.gPos - pg;
                if (r0 <= prev_r0 || sd.gPos <= prev_g0 || rg < 0 || gg < 0) { plain = false; break; }
                if (rg > 0 || gg > 0) { g.rPos = pr; g.gPos = pg; g.rLen = rg; g.gLen = (int)gg; f[w++] = g; }
            }
            Frag x; x.gPos = sd.gPos; x.rPos = r0; x.rLen = x.gLen = sd.len;
            x.ops_off = 0; x.ops_len = 0; x.kind = kSimple; x.meta = 0;
            f[w++] = x;
            prev_r0 = r0; prev_g0 = sd.gPos; pr = r0 + sd.len; pg = sd.gPos + sd.len;
        }
        if (plain && n > 0) {
            if (pr < rlen) { g.rPos = pr; g.gPos = pg; g.rLen = g.gLen = rlen - pr; f[w++] = g; }
            total = w;
        }
    }
    if (total == 0) {
    for (int i = 0; i < n; i++) {
        Frag x; x.gPos = seeds[i].gPos; x.rPos = seeds[i].rPos; x.rLen = x.gLen = seeds[i].len;
        x.ops_off = 0; x.ops_len = 0; x.kind = kSimple; x.meta = 0;
        int j = i - 1;
        while (j >= 0 && frag_before(x, f[j])) { f[j + 1] = f[j]; j--; } // sort by (rPos, gPos), :317
        f[j + 1] = x;
    }
    // RemoveOverlaps (:38-65) then RemoveNullFragPairs (:29-36)
    bool any = false;
    for (int i = 0, j = 1; j < n; i++, j++) {
        if (f[i].rPos == f[j].rPos) { any = true; f[i].rLen = f[i].gLen = 0; }
        else if (f[i].gPos >= f[j].gPos || f[i].gPos + f[i].gLen > f[j].gPos) {
            any = true;
            int ov = (int)(f[i].gPos + f[i].gLen - f[j].gPos);
            if ((f[i].rLen -= ov) < 0) f[i].rLen = 0;
            if ((f[i].gLen -= ov) < 0) f[i].gLen = 0;
        }
    }
    if (any) { int m = 0; for (int i = 0; i < n; i++) if (f[i].rLen != 0) f[m++] = f[i]; n = m; }
    if (n == 0) return -1; // cannot happen for real seeds; treated as invalid
    // IdentifyNormalPairs (:67-108): gap fragments between seeds and at both read ends.
    // The reference appends and merges; gaps always sort between their neighbours, so the
    // list is expanded in place from the back.
    int gaps = 0;
    for (int i = 0; i + 1 < n; i++) {
        int rg = f[i + 1].rPos - (f[i].rPos + f[i].rLen);
        int64_t gg = f[i + 1].gPos - (f[i].gPos + f[i].gLen);
        if (rg > 0 || gg > 0) gaps++;
    }
    bool head = f[0].rPos > 0;
    bool tail = f[n - 1].rPos + f[n - 1].rLen < rlen;
    total = n + gaps + (head ? 1 : 0) + (tail ? 1 : 0);
    int w = total - 1;
    Frag g; g.ops_off = 0; g.ops_len = 0; g.kind = kPlain; g.meta = 0;
    if (tail) {
        g.rPos = f[n - 1].rPos + f[n - 1].rLen; g.gPos = f[n - 1].gPos + f[n - 1].gLen;
        g.rLen = g.gLen = rlen - g.rPos;
        f[w--] = g;
    }
    for (int i = n - 1; i >= 0; i--) {
        Frag cur = f[i];
        f[w--] = cur;
        if (i > 0) {
            int rg = cur.rPos - (f[i - 1].rPos + f[i - 1].rLen); if (rg < 0) rg = 0;
            int64_t gg = cur.gPos - (f[i - 1].gPos + f[i - 1].gLen); if (gg < 0) gg = 0;
            if (rg > 0 || gg > 0) {
                g.rPos = f[i - 1].rPos + f[i - 1].rLen; g.gPos = f[i - 1].gPos + f[i - 1].gLen;
                g.rLen = rg; g.gLen = (int)gg;
                f[w--] = g;
            }
        }
    }
    if (head) {
        Frag &first = f[1];
        g.rPos = 0; g.gPos = first.gPos - first.rPos; g.rLen = g.gLen = first.rPos;
        f[0] = g;
    }
    } // general path
    // CheckAlignmentValidity
    const Frag &a = f[0], &b = f[total - 1];
    if (a.gPos < 0 || b.gPos + b.gLen > ix.G2) return -1;
    int e1 = end_slot(ix, a.gPos), e2 = end_slot(ix, b.gPos + b.gLen - 1);
    if (e1 < 0 || e2 < 0 || ix.end_pos[e1] != ix.end_pos[e2]) return -1;
    return total;
}

// read / genome code of alignment-string position x of a fragment.  Reverse-strand fragments
// (gPos >= G) have both strings reverse-complemented by the reference (ReadAlignment.cpp:179-183);
// complementing both sides does not change any comparison, so only the reversal is applied.
static inline MCX_HD int frag_read_code(const Frag &f, const ReadRef &rd, bool rev, int x)
{
    return read_code(rd, rev ? f.rPos + f.rLen - 1 - x : f.rPos + x);
}
static inline MCX_HD int frag_ref_code(const IndexView &ix, const Frag &f, bool rev, int y)
{
    return ref_code(ix, rev ? f.gPos + f.gLen - 1 - y : f.gPos + y);
}

// bases of a gap fragment with rLen == gLen where read and genome differ (an N on either side differs).  The count does
// not depend on the direction the reference walks the strings in.  A read that carries its 2-bit words (no N) is compared
// sixteen bases at a time: the read's words against a funnel shift of the 2-bit genome.
static inline MCX_HD int frag_mismatches(const IndexView &ix, const Frag &x, const ReadRef &rd)
{
    const int n = x.rLen;
    int mm = 0;
    if (rd.codes) {
        for (int k = 0; k < n; k += 16) {
            const int p = x.rPos + k, w = p >> 4, sh = (p & 15) * 2;
            const uint32_t hi = rd.codes[w], lo = (p & 15) ? rd.codes[w + 1] : 0u; // (a read's words are followed by one more of the slice: never out of bounds)
            const uint32_t rw = sh ? (hi << sh) | (lo >> (32 - sh)) : hi;
            uint32_t d = rw ^ ref_codes16(ix, x.gPos + k);
            d = (d | (d >> 1)) & 0x55555555u;
            const int left = n - k;
            if (left < 16) d &= ~0u << (2 * (16 - left));
            mm += __builtin_popcount(d);
        }
        return mm;
    }
    const bool rev = x.gPos >= ix.G;
    for (int k = 0; k < n; k++)
        if (read_code(rd, rev ? x.rPos + n - 1 - k : x.rPos + k) != ref_code(ix, rev ? x.gPos + n - 1 - k : x.gPos + k)) mm++;
    return mm;
}

struct JobSink {
    DpJob *jobs;
    uint32_t *count;
    uint32_t cap;
};

// size class of a DP problem: 0 = 16-lane groups (target <= 16, query <= 32), 1/2/3 = one wave
// with 1 / 4 / 16 target columns per lane (target <= 64 / 256 / 1024); -1 = refused
static inline MCX_HD int dp_class(int rLen, int gLen)
{
    if (gLen > 1024 || rLen > 2048) return -1;
    if (gLen <= 16 && rLen <= 32) return 0;
    return gLen <= 64 ? 1 : (gLen <= 256 ? 2 : 3);
}

// the k-th DP problem of a pair, from its local list (written by stage_build)
static inline MCX_HD DpJob pair_job(const Ctx &cx, int64_t pair, int k)
{
    PairState st = pair_state(cx.state, cx.lay, cx.caps, pair);
    const int32_t *jl = (const int32_t *)((const uint8_t *)st.hdr + cx.lay.off_jobs);
    const Frag &x = st.frags[jl[2 * k]];
    DpJob j;
    j.pair = (uint32_t)pair; j.slot = (uint16_t)jl[2 * k + 1]; j.rev = x.gPos >= cx.ix.G ? 1 : 0;
    j.rPos = x.rPos; j.rLen = x.rLen; j.gPos = x.gPos; j.gLen = x.gLen;
    j.ops_off = x.ops_off; j.frag = jl[2 * k]; j.score = 0;
    return j;
}

// returns the number of DP problems the pair needs (left in its local list)
// (flags_out: the pair's flags as the stage leaves them, for a caller that lists the pairs that ran over)
static inline MCX_HD int stage_build(const Ctx &cx, int64_t pair, const ReadRef *rd, uint32_t *flags_out = nullptr)
{
    PairState st = pair_state(cx.state, cx.lay, cx.caps, pair);
    PairHdr *const g_hdr = st.hdr;
    PairHdr h = *g_hdr; // in registers through the stage
    h.n_frags = 0; h.n_ops = 0; h.n_jobs = 0;
    // every exit stores back what the stage changes — three words and a short — not the whole header (the unchanged rest would have to
    // sit somewhere for the length of the stage: the compiler parked it in scratch memory)
    auto put_hdr = [&]() { g_hdr->flags = h.flags; g_hdr->n_frags = h.n_frags; g_hdr->n_ops = h.n_ops; g_hdr->n_jobs = h.n_jobs; };
    if (h.flags & kOvAny) { put_hdr(); if (flags_out) *flags_out = h.flags; return 0; }
    int nr = cx.pm.paired ? 2 : 1;
    if (cx.pm.paired) {
        if (h.n_paired == 0) { keep_top_scores(st.cands[0], h.n_cands[0]); keep_top_scores(st.cands[1], h.n_cands[1]); }
        else mask_unpaired(st.cands[0], h.n_cands[0], st.cands[1], h.n_cands[1]);
    } else keep_top_scores(st.cands[0], h.n_cands[0]);

    // local list of DP jobs of this pair (flushed to the global sink at the end)
    int32_t *jl = (int32_t *)((uint8_t *)st.hdr + cx.lay.off_jobs);
    int nj = 0;
    MCX_UNROLL // (written to unroll: a header / read pair indexed by a run-time mate number lives in scratch memory)
    for (int s = 0; s < 2; s++) {
        if (s >= nr) continue;
        Cand *cs = st.cands[s];
        for (int ci = 0; ci < h.n_cands[s]; ci++) {
            const Cand c = cs[ci]; // read once; what changes is stored once (frag_off and n_frags share a word)
            if (c.score == 0) { cs[ci].frag_off = (int16_t)h.n_frags; cs[ci].n_frags = 0; continue; }
            if (h.n_frags + 2 * c.count + 2 > cx.caps.frag_cap) { cs[ci].frag_off = (int16_t)h.n_frags; cs[ci].n_frags = 0; h.flags |= kOvFrags; put_hdr(); if (flags_out) *flags_out = h.flags; return 0; }
            Frag *f = st.frags + h.n_frags;
            int nf = build_frags(cx.ix, rd[s].rlen, c.in_pool ? cx.seed_pool + c.pool_off : st.hits[s] + c.first, c.count, f);
            cs[ci].frag_off = (int16_t)h.n_frags; cs[ci].n_frags = (int16_t)(nf < 0 ? 0 : nf);
            if (nf < 0) { cs[ci].score = 0; continue; }
            // ProcessNormalPair (:155-191): classify each gap fragment
            for (int i = 0; i < nf; i++) {
                Frag x = f[i]; // worked on in registers, stored back once
                if (x.kind == kSimple) continue;
                if (x.rLen > 0 && x.gLen > 0) {
                    bool dp = x.rLen != x.gLen;
                    int mm = -1;
                    if (!dp) {
                        mm = frag_mismatches(cx.ix, x, rd[s]);
                        dp = mm > 1 && mm >= (int)(x.rLen * 0.2);
                    }
                    if (dp) {
                        const int at = ((h.n_ops + 7) & ~7) + kDpSum; // the columns' area, behind the room for their DpSummary
                        if (at + x.rLen + x.gLen > cx.caps.ops_cap) { h.flags |= kOvOps; put_hdr(); if (flags_out) *flags_out = h.flags; return 0; }
                        if (nj >= cx.caps.job_cap) { h.flags |= kOvJobs; put_hdr(); if (flags_out) *flags_out = h.flags; return 0; }
                        x.kind = kDp; x.ops_off = at; x.ops_len = 0; x.meta = 0;
                        h.n_ops = at + x.rLen + x.gLen;
                        jl[2 * nj] = h.n_frags + i; jl[2 * nj + 1] = s;
                        nj++;
                    } else { x.kind = kPlain; x.ops_len = x.rLen; x.meta = (uint32_t)(mm + 1); } // (the count travels with the fragment: the finish stage needs it again)
                } else if (x.rLen > 0) { x.kind = kIns; x.ops_len = x.rLen; }
                else { x.kind = kDel; x.ops_len = x.gLen; }
                f[i] = x;
            }
            h.n_frags += nf;
        }
    }
    h.n_jobs = nj;
    put_hdr();
    if (flags_out) *flags_out = h.flags;
    return nj;
}

// ------------------------------------------------------------------------------------------
// stage G2: post-DP gates, scores, best/sub-best, pair statistics, flags, MAPQ, CIGAR
// ------------------------------------------------------------------------------------------
// column x of a gap fragment: 'M' (both bases), 'I' (read base vs '-'), 'D' ('-' vs genome base)
static inline MCX_HD uint8_t frag_op(const Frag &f, const uint8_t *ops, int x)
{
    switch (f.kind) {
    case kPlain: return 'M';
    case kIns: return 'I';
    case kDel: return 'D';
    default: return ops[f.ops_off + x];
    }
}

// RemoveHeadingGaps (:264-283) / RemoveTailingGaps (:285-304)
static inline MCX_HD bool strip_end_gaps(Frag &f, const uint8_t *ops, bool leading, bool move_pos) // true: the fragment changed
{
    if (f.kind != kDp) return false; // only DP results can start or end with a gap column
    int rs = 0, gs = 0, j = 0;
    if (f.meta) { // counted by the DP kernel
        const DpSummary &sm = *(const DpSummary *)(ops + ((int)f.meta - 1) * 8);
        rs = leading ? sm.lead_i : sm.tail_i; gs = leading ? sm.lead_d : sm.tail_d; j = rs + gs;
        if (j > 0) { if (leading) f.ops_off += j; f.ops_len -= j; }
    } else if (leading) {
        for (; j < f.ops_len; j++) { uint8_t o = ops[f.ops_off + j]; if (o == 'D') gs++; else if (o == 'I') rs++; else break; }
        if (j > 0) { f.ops_off += j; f.ops_len -= j; }
    } else {
        for (; j < f.ops_len; j++) { uint8_t o = ops[f.ops_off + f.ops_len - 1 - j]; if (o == 'D') gs++; else if (o == 'I') rs++; else break; }
        if (j > 0) f.ops_len -= j;
    }
    if (j > 0) {
        f.rLen -= rs; f.gLen -= gs;
        if (move_pos) { f.rPos += rs; f.gPos += gs; }
    }
    return j > 0;
}

struct ColStats { int switches, n, mis, match; };

// one pass over the columns of a gap fragment: CheckLocalAlignmentQuality (:193-232),
// EvaluateAlignmentScore (:234-245) and FindMisMatchNumber (:247-262) all read from it
static inline MCX_HD ColStats frag_columns(const IndexView &ix, const Frag &f, const uint8_t *ops, const ReadRef &rd)
{
    ColStats cs; cs.switches = cs.n = cs.mis = cs.match = 0;
    if (f.kind != kDp) { // one kind of column throughout: no walk
        if (f.ops_len > 0) cs.switches = 1;
        if (f.kind == kPlain) { cs.n = f.ops_len; cs.mis = f.meta ? (int)f.meta - 1 : frag_mismatches(ix, f, rd); cs.match = cs.n - cs.mis; }
        return cs;
    }
    if (f.meta) { // counted by the DP kernel; an end the gates trimmed takes its runs of gap columns with it
        const DpSummary &sm = *(const DpSummary *)(ops + ((int)f.meta - 1) * 8);
        cs.n = sm.n; cs.mis = sm.mis; cs.match = cs.n - cs.mis;
        cs.switches = sm.switches - ((uint32_t)f.ops_off != sm.cols_off ? sm.lead_runs : 0)
                                  - ((uint32_t)(f.ops_off + f.ops_len) != sm.cols_off + sm.cols_len ? sm.tail_runs : 0);
        return cs;
    }
    bool rev = f.gPos >= ix.G;
    int kind = -1, ri = 0, gi = 0;
    for (int x = 0; x < f.ops_len; x++) {
        uint8_t o = frag_op(f, ops, x);
        int k;
        if (o == 'D') { k = 0; gi++; }
        else if (o == 'I') { k = 1; ri++; }
        else {
            k = 2; cs.n++;
            if (frag_read_code(f, rd, rev, ri) != frag_ref_code(ix, f, rev, gi)) cs.mis++; else cs.match++;
            ri++; gi++;
        }
        if (k != kind) { kind = k; cs.switches++; }
    }
    return cs;
}

static inline MCX_HD bool quality_ok(const ColStats &c)
{
    return !(c.switches >= 4 || (c.mis >= 3 && c.mis >= (int)(c.n * 0.3)));
}

// Fragment i of a candidate in alignment order.  The reference reverses the fragment vector of a
// reverse-strand candidate (ReadAlignment.cpp:412-416); here the list stays as built and is read backwards
// (the swaps were two stores per pair of fragments, and stores are what the finish stage is short of).
static inline MCX_HD int frag_index(const Cand &c, int i) { return c.frag_off + (c.fwd ? i : c.n_frags - 1 - i); }

// ProduceReadAlignment from :336 on, for one read
// (ix: the view the read's genome bases are fetched through — cx.ix, or one whose pac points into a window held in LDS)
static inline MCX_HD void extend_read(const Ctx &cx, const IndexView &ix, PairState &st, int s, const ReadRef &rd)
{
    PairHdr &h = *st.hdr;
    ReadSum sum = h.sum[s];
    const int max_mm = (int)(rd.rlen * cx.pm.max_mm_rate);
    const int min_score = (int)(rd.rlen * (1 - cx.pm.max_mm_rate));
    Cand *cs = st.cands[s];
    const int n_cands = h.n_cands[s];
    // structures are copied to registers, worked on, and stored back whole: a lane's memory
    // instructions, not its bytes, are what this stage is short of
    for (int ci = 0; ci < n_cands; ci++) {
        Cand c = cs[ci];
        if (c.score == 0) continue;
        Frag *f = st.frags + c.frag_off;
        const int num = c.n_frags, last = num - 1;
        bool head_ok = true, tail_ok = true;
        int score = 0, mism = 0;
        bool dead = false;
        int64_t g_first = 0; // gPos of the first fragment after the loop
        for (int i = 0; i < num; i++) {
            Frag x = f[i];
            if (i == 0) g_first = x.gPos;
            if (x.kind == kSimple) { score += x.rLen; continue; }
            const bool fwd = x.gPos < ix.G;
            if (i == 0) {
                bool changed = strip_end_gaps(x, st.ops, fwd, true);
                ColStats q = frag_columns(ix, x, st.ops, rd);
                if (x.ops_len >= kMinAlnBlockSize && !quality_ok(q)) {
                    head_ok = false;
                    const Frag nx = f[i + 1];
                    x.rLen = x.gLen = 0; x.ops_len = 0; x.kind = kEmpty;
                    x.rPos = nx.rPos; x.gPos = nx.gPos;
                    changed = true;
                } else { score += q.match; mism += q.mis; }
                g_first = x.gPos;
                if (changed) f[i] = x; // (stored only when it changed: stores are what this stage is short of)
            } else if (i == last) {
                bool changed = strip_end_gaps(x, st.ops, !fwd, false);
                ColStats q = frag_columns(ix, x, st.ops, rd);
                if (x.ops_len >= kMinAlnBlockSize && !quality_ok(q)) {
                    tail_ok = false;
                    const Frag pv = f[i - 1];
                    x.rLen = x.gLen = 0; x.ops_len = 0; x.kind = kEmpty;
                    x.rPos = pv.rPos + pv.rLen; x.gPos = pv.gPos + pv.gLen;
                    changed = true;
                } else { score += q.match; mism += q.mis; }
                if (changed) f[i] = x;
            } else {
                ColStats q = frag_columns(ix, x, st.ops, rd);
                if (x.rLen >= kMinAlnBlockSize && x.gLen >= kMinAlnBlockSize && !quality_ok(q)) { dead = true; break; }
                score += q.match; mism += q.mis;
            }
        }
        if (dead || (!head_ok && !tail_ok)) { cs[ci].score = 0; continue; }
        if (score == 0 || (score < min_score && mism > max_mm)) { cs[ci].score = 0; continue; }
        const int8_t fwd = g_first < ix.G ? 1 : 0;
        if (score != c.score) cs[ci].score = score; // (only the fields that changed)
        if (fwd != c.fwd) cs[ci].fwd = fwd;
        if (score > sum.score) { sum.score = score; sum.best = ci; }
        else if (score > sum.sub) sum.sub = score;
    }
    if (n_cands > 1) for (int ci = 0; ci < n_cands; ci++) if (cs[ci].score < sum.score) cs[ci].score = 0;
    h.sum[s] = sum;
}

struct Coord { int64_t pos; int32_t chr; };

static inline MCX_HD Coord to_coord(const IndexView &ix, int64_t g) // DetermineCoordinate, tools.cpp:132-164
{
    Coord c;
    if (g < ix.G) {
        if (ix.n_chr == 1) { c.chr = 0; c.pos = g + 1; }
        else { int s = end_slot(ix, g); c.chr = ix.end_chr[s]; c.pos = g + 1 - ix.chr_fwd[c.chr]; }
    } else {
        if (ix.n_chr == 1) { c.chr = 0; c.pos = ix.G2 - g; }
        else { int s = end_slot(ix, g); c.chr = ix.end_chr[s]; c.pos = ix.end_pos[s] - g + 1; }
    }
    return c;
}

static inline MCX_HD Coord aln_coord(const IndexView &ix, const Cand &c, const Frag *frags) // GetAlnCoordinate, SamReport.cpp:121-149
{
    Coord k; k.pos = 0; k.chr = 0;
    for (int i = 0; i < c.n_frags; i++) {
        const Frag x = frags[frag_index(c, i)];
        if (x.gLen > 0) return to_coord(ix, c.fwd ? x.gPos : x.gPos + x.gLen - 1);
    }
    return k;
}

static inline MCX_HD int mapq_of(const Ctx &cx, const ReadSum &r) // EvaluateMAPQ, SamReport.cpp:86-101
{
    if (r.score == 0 || r.score == r.sub) return 0;
    if (r.sub == 0 || r.score - r.sub > 5) return 60;
    int row = r.score < cx.mapq_rows ? r.score : cx.mapq_rows - 1;
    return cx.mapq_tab[row * 6 + (r.score - r.sub)];
}

// GenerateCIGARstring (SamReport.cpp:172-316) as BAM-style (len << 4 | op) words; op codes
// M=0 I=1 D=2 S=4.  Returns the number of words needed (may exceed cap).
// (out may be null with cap 0: the operations are only counted)
// (stride: operation k goes to out[k * stride] — the finish kernel stages them word-major in LDS, one lane beside the next)
static inline MCX_HD int cigar_of(int rlen, const Cand &c, const Frag *frags, const uint8_t *ops, uint32_t *out, int cap, int stride = 1)
{
    auto v = [&](int i) -> Frag { return frags[frag_index(c, i)]; };
    int num = c.n_frags, n = 0, run = 0, st = -1;
    auto put = [&](int len, int op) {
        const uint32_t w = ((uint32_t)len << 4) | (uint32_t)op;
        if (n < cap) out[n * stride] = w;
        n++;
    };
    auto flush_to = [&](int ns) { if (st != ns) { if (run > 0) put(run, st); st = ns; run = 0; } };
    {
        const Frag f0 = v(0);
        if (f0.kind != kSimple) {
            int clip = c.fwd ? f0.rPos : rlen - (f0.rPos + f0.rLen);
            if (clip > 0) put(clip, 4);
        }
    }
    for (int i = 0; i < num; i++) {
        const Frag f = v(i); // a copy: the stores of the operations below could alias a reference
        if (f.kind == kSimple) { flush_to(0); run += f.rLen; }
        else if (f.kind == kEmpty) continue;
        else if (f.ops_len > 0) {
            if (f.kind != kDp) { flush_to(f.kind == kDel ? 2 : (f.kind == kIns ? 1 : 0)); run += f.ops_len; } // one kind of column throughout
            else if (f.meta && ((const DpSummary *)(ops + ((int)f.meta - 1) * 8))->n_rle != 0xFFFF) { // the runs the DP kernel left, less the trimmed ends
                const DpSummary &sm = *(const DpSummary *)(ops + ((int)f.meta - 1) * 8);
                const int k0 = kDpRle - sm.n_rle + ((uint32_t)f.ops_off != sm.cols_off ? sm.lead_runs : 0);
                const int k1 = kDpRle - ((uint32_t)(f.ops_off + f.ops_len) != sm.cols_off + sm.cols_len ? sm.tail_runs : 0);
                for (int k = k0; k < k1; k++) { const uint32_t e = sm.rle[k]; flush_to((int)(e & 15u)); run += (int)(e >> 4); }
            }
            else for (int x = 0; x < f.ops_len; x++) {
                const uint8_t o = ops[f.ops_off + x];
                flush_to(o == 'D' ? 2 : (o == 'I' ? 1 : 0));
                run++;
            }
        } else if (f.rLen > 0) { flush_to(1); run += f.rLen; }
        else if (f.gLen > 0) { flush_to(2); run += f.gLen; }
    }
    if (run > 0) put(run, st);
    if (num > 1) {
        const Frag fl = v(num - 1);
        if (fl.kind != kSimple) {
            int clip = c.fwd ? rlen - (fl.rPos + fl.rLen) : fl.rPos;
            if (clip > 0) put(clip, 4);
        }
    }
    return n;
}

// SetPairedAlignmentFlag (SamReport.cpp:26-84) for one candidate
static inline MCX_HD int paired_flag(const Cand &c, const Cand *other, bool first, bool unique_branch)
{
    int fl = first ? 0x41 : 0x81;
    if (first) fl |= c.fwd ? 0x20 : 0x10; else fl |= c.fwd ? 0x10 : 0x20;
    if (c.mate != -1 && other[c.mate].score > 0) fl |= 0x2;
    else {
        if (unique_branch) { if (first) fl |= c.fwd ? 0x10 : 0x20; else fl |= c.fwd ? 0x20 : 0x10; }
        fl |= 0x8;
    }
    return fl;
}

// GenCoordinatePair (ReadMapping.cpp:361-394) + the counting at :479-531
static inline MCX_HD void pair_stats(const Ctx &cx, PairState &st, DetailHdr *dh)
{
    PairHdr &h = *st.hdr;
    const Cand *c1 = st.cands[0], *c2 = st.cands[1];
    int n1 = h.n_cands[0], n2 = h.n_cands[1];
    int64_t dist = 0, g1 = 0, g2 = 0;
    for (int i = 0; i < n1; i++) {
        const Cand &c = c1[i];
        if (c.score > 0 && c.mate != -1 && c2[c.mate].score > 0) {
            g1 = st.frags[frag_index(c, 0)].gPos; g2 = st.frags[frag_index(c2[c.mate], 0)].gPos;
            dist = g2 > g1 ? g2 - g1 : g1 - g2;
            break;
        }
    }
    if (dist == 0) {
        int a = 0, b = 0;
        int64_t ga = 0, gb = 0;
        for (int i = 0; i < n1; i++) if (c1[i].score > 0) { if (a == 0) ga = st.frags[frag_index(c1[i], 0)].gPos; a++; }
        for (int i = 0; i < n2; i++) if (c2[i].score > 0) { if (b == 0) gb = st.frags[frag_index(c2[i], 0)].gPos; b++; }
        if (a == 1 && b == 1) { g1 = ga; g2 = gb; dist = g2 > g1 ? g2 - g1 : g1 - g2; }
        else if (a == 0 && b >= 1) { g1 = -1; dist = g2 = gb; }
        else if (a >= 1 && b == 0) { dist = g1 = ga; g2 = -1; }
    }
    h.pair_ok = 0; h.pair_dist = 0;
    if (dh) { dh->disc_kind = 0; dh->disc_g1 = g1; dh->disc_g2 = g2; dh->disc_dist = dist; }
    if (dist != 0 && g1 != -1 && g2 != -1) {
        const int64_t G = cx.ix.G;
        bool inv = (g1 < G && g2 >= G) || (g1 >= G && g2 < G);
        if (!inv && dist <= kMinTranslocationSize) { h.pair_ok = 1; h.pair_dist = (int)dist; }
        if (dh) { // which branch of ReadMapping.cpp:486-521 the pair takes when -vcf is on
            if (g1 < G && g2 >= G) dh->disc_kind = 1;
            else if (g1 >= G && g2 < G) dh->disc_kind = 2;
            else if (dist > kMinTranslocationSize) dh->disc_kind = (g1 < G && g2 < G) ? 3 : 4;
        }
    }
}

// one output record: the line GeneratePairedSamStream / GenerateSingleSamStream print for this
// read in unique mode (SamReport.cpp:324-488).  cig: where the read's n_cig operations go (counted by
// finish_scores, reserved in the batch's pool by the caller), cig_off: that place as a pool offset.
static inline MCX_HD void emit_record(const Ctx &cx, PairState &st, int s, const ReadRef *rd, AlnRec &dst,
                                      uint32_t *cig, int n_cig, uint32_t cig_off, const uint32_t *staged = nullptr, int stage_stride = 1)
{
    PairHdr &h = *st.hdr;
    const ReadSum me = h.sum[s];
    AlnRec out;
    out.pos = 0; out.mate_pos = 0; out.chr = -1; out.flag = 0; out.mapq = 0; out.tlen = 0;
    out.nm = 0; out.as = 0; out.xs = 0; out.n_cigar = 0; out.fwd = 1; out.has_mate = 0; out.pad[0] = out.pad[1] = 0;
    const bool paired = cx.pm.paired != 0;
    if (me.score == 0) {
        if (!paired) { out.flag = 4; dst = out; return; }
        const ReadSum &ot = h.sum[1 - s];
        int fl = 0x1 | 0x4 | (s == 0 ? 0x40 : 0x80);
        if (ot.score == 0) fl |= 0x8;
        else if (h.n_cands[1 - s] > 0) fl |= 0x30; // both strand bits, SamReport.cpp:401-402 / :449-450
        out.flag = fl;
        dst = out;
        return;
    }
    Cand c = st.cands[s][me.best];
    if (paired) {
        const Cand *oc = st.cands[1 - s];
        // flags are set for every surviving candidate when the best score is tied; the line
        // printed in unique mode is the first one, candidate `best`
        c.flag = paired_flag(c, oc, s == 0, me.score > me.sub);
    } else c.flag = c.fwd ? 0 : 0x10; // SetSingledAlignmentFlag, SamReport.cpp:7-24
    out.flag = c.flag;
    out.mapq = mapq_of(cx, me);
    Coord km = aln_coord(cx.ix, c, st.frags);
    out.chr = km.chr; out.pos = km.pos;
    out.fwd = c.fwd;
    out.nm = rd[s].rlen - c.score; out.as = me.score; out.xs = me.sub;
    if (!cig) out.n_cigar = 0; // (the pool ran over; the batch fails)
    else if (staged && n_cig <= kCigStage) { for (int k = 0; k < n_cig; k++) cig[k] = staged[k * stage_stride]; out.n_cigar = n_cig; }
    else out.n_cigar = cigar_of(rd[s].rlen, c, st.frags, st.ops, cig, n_cig);
    out.pad[0] = (int32_t)cig_off;
    if (paired) {
        const ReadSum &ot = h.sum[1 - s];
        int j = c.mate;
        if (j != -1 && ot.score > 0 && st.cands[1 - s][j].score == ot.score) {
            const Cand &o = st.cands[1 - s][j];
            Coord ko = aln_coord(cx.ix, o, st.frags);
            // TLEN is defined from read 1's side and negated for read 2 (SamReport.cpp:428, :475)
            const Cand &cand1 = s == 0 ? c : o;
            int64_t p1 = s == 0 ? km.pos : ko.pos, p2 = s == 0 ? ko.pos : km.pos;
            int dist = (int)(p2 - p1 + (cand1.fwd ? rd[1].rlen : 0 - rd[0].rlen));
            out.tlen = s == 0 ? dist : 0 - dist;
            out.mate_pos = ko.pos;
            out.has_mate = 1;
        }
    }
    dst = out;
}

// copies what the profile stage needs of one read out of the pair state (which is reused by the
// next tier / replay pass): the fragments and DP columns of its single surviving candidate, or
// the genome ranges of all of them when it is multi-mapped (ReadMapping.cpp:566-571)
static inline MCX_HD void write_detail(const Ctx &cx, PairState &st, int s, uint8_t *rec)
{
    PairHdr &h = *st.hdr;
    DetailHdr &d = *(DetailHdr *)rec;
    Frag *df = (Frag *)(rec + sizeof(DetailHdr));
    uint8_t *dops = rec + cx.dlay.off_ops;
    d.type = 0; d.n_frags = 0; d.fwd = 1; d.n_ops = 0; d.frag0 = 0;
    if (h.sum[s].score == 0) return;
    const Cand *cs = st.cands[s];
    int live = 0, one = -1;
    for (int i = 0; i < h.n_cands[s]; i++) if (cs[i].score > 0) { live++; one = i; }
    if (live == 1) {
        const Cand &c = cs[one];
        if (c.n_frags > cx.dlay.frag_cap) { h.flags |= kOvDetail; return; }
        d.type = 1; d.fwd = c.fwd; d.n_frags = c.n_frags;
        int no = 0;
        for (int i = 0; i < c.n_frags; i++) {
            Frag f = st.frags[frag_index(c, i)];
            if (f.kind == kDp) {
                if (no + f.ops_len > cx.dlay.ops_cap) { h.flags |= kOvDetail; d.type = 0; return; }
                for (int x = 0; x < f.ops_len; x++) dops[no + x] = st.ops[f.ops_off + x];
                f.ops_off = no; no += f.ops_len;
            }
            df[i] = f;
        }
        d.n_ops = no;
    } else {
        d.type = 2;
        int n = 0;
        for (int i = 0; i < h.n_cands[s]; i++) {
            const Cand &c = cs[i];
            if (c.score <= 0) continue;
            if (n >= cx.dlay.frag_cap) { h.flags |= kOvDetail; d.type = 0; return; }
            const Frag &a = st.frags[frag_index(c, 0)], &b = st.frags[frag_index(c, c.n_frags - 1)];
            const int64_t g0 = c.fwd ? a.gPos : cx.ix.G2 - (a.gPos + a.gLen);
            const int64_t g1 = c.fwd ? b.gPos + b.gLen : cx.ix.G2 - b.gPos;
            Frag r; r.gPos = g0; r.rPos = 0; r.rLen = (int32_t)(g1 - g0); r.gLen = 0; r.ops_off = 0; r.ops_len = 0; r.kind = kSimple; r.meta = 0;
            df[n++] = r;
        }
        d.n_frags = n;
    }
}

// Stage G2 comes in two steps so that a wavefront can reserve the CIGAR words of its 64 pairs with one atomic in
// between.  st.hdr points at the caller's copy of the header (registers on the device).
// finish_scores: gates, scores, best / sub-best, pair statistics; n_cig[s] = CIGAR operations read s will print.
// (stage: room for kCigStage operations per read — most reads have at most that many, and finish_records then copies them
//  instead of walking the alignment a second time)
static inline MCX_HD void finish_scores(const Ctx &cx, PairState &st, const ReadRef *rd, DetailHdr *dh, int n_cig[2], const IndexView *ixr = nullptr,
                                        uint32_t *stage = nullptr, int stage_stride = 1)
{
    PairHdr &h = *st.hdr;
    n_cig[0] = n_cig[1] = 0;
    if (h.flags & kOvAny) return;
    const int nr = cx.pm.paired ? 2 : 1;
    h.mapped = 0;
    MCX_UNROLL
    for (int s = 0; s < 2; s++) { if (s >= nr) break; extend_read(cx, ixr ? ixr[s] : cx.ix, st, s, rd[s]); if (h.sum[s].score > 0) h.mapped++; }
    if (cx.pm.paired) pair_stats(cx, st, dh); else { h.pair_ok = 0; h.pair_dist = 0; if (dh) dh->disc_kind = 0; }
    MCX_UNROLL
    for (int s = 0; s < 2; s++) {
        if (s >= nr) break;
        if (h.sum[s].score > 0) n_cig[s] = cigar_of(rd[s].rlen, st.cands[s][h.sum[s].best], st.frags, st.ops, stage ? stage + s * kCigStage * stage_stride : nullptr, stage ? kCigStage : 0, stage_stride);
    }
}

// finish_records: the pair's output records (rec2[0..nr)), their CIGAR words at cig_pool + cig_off[s] (null pool: the
// reservation failed), and the alignment detail of its reads when the profile is kept (detail2 = the pair's first record).
static inline MCX_HD void finish_records(const Ctx &cx, PairState &st, const ReadRef *rd, AlnRec *rec2, uint32_t *cig_pool,
                                         const uint32_t cig_off[2], const int n_cig[2], uint8_t *detail2, const uint32_t *stage = nullptr, int stage_stride = 1)
{
    PairHdr &h = *st.hdr;
    if (h.flags & kOvAny) return;
    const int nr = cx.pm.paired ? 2 : 1;
    MCX_UNROLL
    for (int s = 0; s < 2; s++) {
        if (s >= nr) break;
        emit_record(cx, st, s, rd, rec2[s], cig_pool ? cig_pool + cig_off[s] : nullptr, n_cig[s], cig_off[s], stage ? stage + s * kCigStage * stage_stride : nullptr, stage_stride);
        if (detail2) write_detail(cx, st, s, detail2 + (int64_t)s * cx.dlay.stride);
    }
}

// both steps for callers that take the pool's words front to back (host emulation)
static inline MCX_HD void stage_finish(const Ctx &cx, int64_t pair, const ReadRef *rd, AlnRec *recs, uint8_t *detail0, PairHdr *keep = nullptr)
{
    PairState st = pair_state(cx.state, cx.lay, cx.caps, pair);
    PairHdr *const g_hdr = st.hdr;
    PairHdr h = *g_hdr; // the header travels in registers through this stage and is stored back once
    st.hdr = &h;
    const int nr = cx.pm.paired ? 2 : 1;
    DetailHdr *dh = detail0 ? (DetailHdr *)(detail0 + (pair * nr) * cx.dlay.stride) : nullptr;
    int n_cig[2];
    finish_scores(cx, st, rd, dh, n_cig);
    uint32_t off[2] = {*cx.cig_pool_n, *cx.cig_pool_n + (uint32_t)n_cig[0]};
    const bool fits = off[1] + (uint32_t)n_cig[1] <= cx.cig_pool_cap;
    if (fits) *cx.cig_pool_n = off[1] + (uint32_t)n_cig[1];
    finish_records(cx, st, rd, recs + pair * nr, fits ? cx.cig_pool : nullptr, off, n_cig, detail0 ? detail0 + (pair * nr) * cx.dlay.stride : nullptr);
    if (keep) *keep = h; else *g_hdr = h;
}

#if defined(__HIPCC__)
constexpr int kLdsEnds = 1024; // chromosome ends a block keeps in LDS

// Reserves n slots of a work list for every lane of the wave with ONE atomic: an inclusive scan
// over the wave, the last lane adds the total.  Must be reached by all 64 lanes (n = 0 for the
// ones with nothing to append).  The lists' order never influences a result.
static __device__ __forceinline__ uint32_t wave_reserve(uint32_t *counter, uint32_t n)
{
    const int lane = threadIdx.x & 63;
    uint32_t incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    const uint32_t total = __shfl(incl, 63, 64);
    uint32_t base = 0;
    if (lane == 63 && total) base = atomicAdd(counter, total);
    base = __shfl(base, 63, 64);
    return base + incl - n;
}

#endif

} // namespace mcx
#endif
