// mapcaller_amd/csrc/mcx_profile.h — the -vcf bookkeeping of the mapping loop on the GPU (device only).
//
// Replaces UpdateProfile / UpdateMultiHitCount (reference src/AlignmentProfile.cpp:41-271), which
// the reference runs serially under ProfileLock (src/ReadMapping.cpp:562-573) into a 16-byte
// bit-field record per genome position (MappingRecord_t, src/structure.h:152-163).
//
// Here the per-position counters are ten planes (A C G T multi_hit readCount F1 R2 F2 R1) in caller-owned
// HBM — multi_hit 32 bits wide, the other nine 16: mcx_planes.h, 22 bytes per position —, so that a
// multi-GPU run can sum them with plain reduces; the field widths of the reference (12-bit saturation
// at 4095, 16-bit wrap, duplicate cap) are applied afterwards by k_prof_finalize.  One lane
// accumulates one read (k_prof_accum).
//
// Most of what a read adds is a run of +1 over consecutive positions: its strand plane over the
// whole read, the multi-hit plane over a candidate's span, and the base planes under an exact seed
// (where the read's base IS the reference's).  Those runs are written as differences — +1 at the
// first position, -1 behind the last — into the strand / multi planes themselves and into one more
// plane for "read base equals reference base" (ProfView::match, owned by the context), and
// mcx_profile_settle turns the differences into counts once, when the run's batches are all in:
// an inclusive scan per plane (u32 wrap-around makes the -1 exact), then match[p] is added to the
// plane of the reference's base at p.  Two atomics per run instead of one per position.  The only
// order-dependent rule — at most iMaxDuplicate uniquely mapped reads are admitted per start
// position, in input order (AlignmentProfile.cpp:76-77) — is decided before accumulation by
// sorting (start, read index) keys.  Insert / delete strings and break points are sparse
// records appended to a list that the host folds into maps.
#ifndef MCX_PROFILE_H
#define MCX_PROFILE_H
#include "mcx_glue.h"
#include "mcx_planes.h"

namespace mcx {

#if defined(__HIPCC__)

struct ProfView {
    PlanesView pl;    // the ten planes (mcx_planes.h); multi_hit and the strand planes hold differences until mcx_profile_settle
    uint16_t *match;  // [stride] differences of "read base == reference base" coverage
    int64_t G;
    int32_t max_dup, max_clip;
};

struct SparseSink { SparseRec *recs; uint32_t *n; uint32_t cap; uint32_t *refused; };

static __device__ __forceinline__ void sparse_put(const SparseSink &s, const SparseRec &r)
{
    const uint32_t at = atomicAdd(s.n, 1u);
    if (at < s.cap) s.recs[at] = r;
}

// A million tally records a batch, every one a fetch-and-add on the list's one counter, is a queue at one L2 channel: a
// wavefront collects its records in LDS and takes their places in the list with one atomic (wave_sparse_flush, called where
// the wavefront's lanes are together again).  A record that finds the LDS buffer full goes to the list directly.
struct WaveSparse { SparseRec *buf; uint32_t *cnt; uint32_t cap; };

static __device__ __forceinline__ void wave_sparse_put(const WaveSparse &w, const SparseSink &s, const SparseRec &r)
{
    const uint32_t at = atomicAdd(w.cnt, 1u);
    if (at < w.cap) w.buf[at] = r;
    else sparse_put(s, r);
}

static __device__ __forceinline__ void wave_sparse_flush(const WaveSparse &w, const SparseSink &s)
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    const int lane = threadIdx.x & 63;
    const uint32_t n = *w.cnt < w.cap ? *w.cnt : w.cap;
    if (n) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(s.n, n);
        base = (uint32_t)__shfl((int)base, 0, 64);
        const U4 *src = (const U4 *)w.buf;
        for (uint32_t k = lane; k < n * 4; k += 64) // (a record is four 16-byte words)
            if (base + (k >> 2) < s.cap) ((U4 *)(s.recs + base))[k] = src[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    if (lane == 0) *w.cnt = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
}

static __device__ __forceinline__ const DetailHdr &detail_hdr(const uint8_t *detail, const DetailLayout &dl, uint32_t r)
{
    return *(const DetailHdr *)(detail + (uint64_t)r * dl.stride);
}

// pass 1: break points, clip gate, and the (start position, read) key of every read that reaches
// the duplicate check (AlignmentProfile.cpp:53-77); n_valid counts them (the others get ~0 and sort last)
__global__ void k_prof_keys(const uint8_t *detail, DetailLayout dl, ReadBatch rb, IndexView ix, ProfView pv, SparseSink sink,
                            uint64_t *keys, uint32_t *n_valid)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t key = ~0ull;
    if (r < rb.n_reads) {
        const uint8_t *rec = detail + (uint64_t)r * dl.stride;
        const DetailHdr &d = *(const DetailHdr *)rec;
        if (d.type == 1) {
            const Frag *f = (const Frag *)(rec + sizeof(DetailHdr)) + d.frag0;
            const Frag &a = f[0], &b = f[d.n_frags - 1];
            const int rlen = (int)(rb.off[r + 1] - rb.off[r]);
            bool go = true;
            if (a.rLen == 0 && a.gLen == 0) {
                if (a.rPos > 20) { SparseRec s; s.pos = a.gPos < ix.G ? a.gPos : ix.G2 - 1 - a.gPos; s.type = 'B'; s.len = 0; sparse_put(sink, s); }
                if (a.rPos > pv.max_clip) go = false;
            }
            if (go && b.rLen == 0 && b.gLen == 0) {
                if (rlen - b.rPos > 20) { SparseRec s; s.pos = b.gPos < ix.G ? b.gPos : ix.G2 - 1 - b.gPos; s.type = 'B'; s.len = 0; sparse_put(sink, s); }
                if (rlen - b.rPos > pv.max_clip) go = false;
            }
            if (go) {
                const int64_t g = d.fwd ? a.gPos : ix.G2 - (a.gPos + a.gLen);
                key = ((uint64_t)g << 32) | r;
            }
        }
        keys[r] = key;
    }
    const uint64_t m = __ballot(key != ~0ull);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_valid, (uint32_t)__popcll(m));
}

// pass 2 (keys sorted by (start position, read number in input order), the reads of every shard of the
// round among them): a read is admitted when fewer than max_dup reads were admitted at its start position
// before it — earlier rounds (readCount plane) plus earlier reads of this round.  Only the reads
// [own_lo, own_lo + n_own) are this shard's: their flags are written.
// (n_dev: the number of keys where the host has not looked at it — a batch's bookkeeping queued behind its mapping; the grid then covers the batch's reads)
__global__ void k_prof_admit(const uint64_t *keys, uint64_t n, ProfView pv, uint8_t *admit, uint32_t own_lo, uint32_t n_own, const uint32_t *n_dev = nullptr)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = *n_dev;
    if (j >= n) return;
    const uint64_t key = keys[j];
    const uint32_t idx = (uint32_t)key - own_lo;
    if (idx >= n_own) return;
    const uint64_t g = key >> 32;
    int rank = 0;
    for (int k = 1; k <= pv.max_dup && (uint64_t)k <= j; k++) { if ((keys[j - k] >> 32) == g) rank++; else break; }
    const uint32_t before = pv.pl.h(kPlReadCount)[g];
    admit[idx] = (uint8_t)((admit[idx] & 2) | ((before + (uint32_t)rank < (uint32_t)pv.max_dup) ? 1 : 0)); // (bit 1: k_pack_reads')
}

// pass 2b (after every flag is out): the first key of each start position adds the round's admissions to
// readCount — the count of the whole run, the same on every shard (`readCount < iMaxDuplicate` then ++, :76-77)
__global__ void k_prof_count(const uint64_t *keys, uint64_t n, ProfView pv, const uint32_t *n_dev = nullptr)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = *n_dev;
    if (j >= n) return;
    const uint64_t g = keys[j] >> 32;
    if (j > 0 && (keys[j - 1] >> 32) == g) return;
    uint32_t run = 1;
    while (run < (uint32_t)pv.max_dup && j + run < n && (keys[j + run] >> 32) == g) run++;
    uint16_t *cnt = &pv.pl.h(kPlReadCount)[g]; // (a 16-bit store: the neighbouring position's half of the word is another thread's)
    const uint32_t v = (uint32_t)*cnt + run;
    *cnt = (uint16_t)(v < (uint32_t)pv.max_dup ? v : (uint32_t)pv.max_dup);
}

// the character the reference sees at alignment-string index xi of a fragment's read string
static __device__ __forceinline__ uint8_t frag_read_char(const ReadRef &rd, const Frag &f, bool fwd, int xi)
{
    if (fwd) return read_char(rd, f.rPos + xi);
    switch (read_char(rd, f.rPos + f.rLen - 1 - xi)) { // SelfComplementarySeq, ReadAlignment.cpp:181
    case 'A': case 'a': return 'T';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    case 'T': case 't': return 'A';
    default: return 'N';
    }
}

// +1 over [lo, hi) of a plane kept as differences (clipped to the genome): the 32-bit plane, a 16-bit one
static __device__ __forceinline__ void range_add(uint32_t *plane, int64_t lo, int64_t hi, int64_t G)
{
    if (lo < 0) lo = 0;
    if (hi > G) hi = G;
    if (lo >= hi) return;
    atomicAdd(plane + lo, 1u);
    if (hi < G) atomicAdd(plane + hi, 0xFFFFFFFFu);
}
static __device__ __forceinline__ void range_add(uint16_t *plane, int64_t lo, int64_t hi, int64_t G)
{
    if (lo < 0) lo = 0;
    if (hi > G) hi = G;
    if (lo >= hi) return;
    half_inc(plane, (uint64_t)lo);
    if (hi < G) half_dec(plane, (uint64_t)hi);
}

// (which reads hold a byte that is not one of the upper-case letters ACGT — bit 1 of the read's flag byte; bit 0 is the admission — is known
//  since the batch was packed: k_pack_reads, mcx_pipeline.hip)

// pass 3: one read per lane.  Nearly everything a read adds is a handful of runs (its strand plane over the read, the base
// planes under each exact seed): a few scalar decisions per fragment — with a wavefront per read all 64 lanes repeated them
// (the kernel was bound by instruction issue: 660 wave instructions a read).  The fragments whose columns have to be walked
// one by one (gaps between seeds, DP fragments, inserts and deletions) go to a list, with everything the walk needs, and
// k_prof_cols walks them sixteen columns at a time; a gap of up to four bases (the usual one: a single substitution) is
// done on the spot, and so is everything once the list is full.
struct ColItem {
    Frag f;
    uint32_t read, off;     // read number in the batch; where its bases start
    int32_t rlen;
    uint32_t flags;         // 1: forward candidate, 2: the read is a flipped mate 2
};
static_assert(sizeof(ColItem) == 32, "ColItem is two 16-byte words");
struct ColList { ColItem *items; uint32_t *n; uint32_t cap; };

// the columns [0, n_cols) of one fragment, lanes `me`, `me + width`, .. of a group of `width` lanes that all call this
// (width 1: a lane by itself).  `grp_shift`: position of the group's lanes in the wavefront's ballots.
template <int WIDTH>
static __device__ __forceinline__ void prof_walk(const ReadRef &rd, const Frag &f, bool fwd, const uint8_t *ops, const IndexView &ix,
                                                 const ProfView &pv, const SparseSink &sink, const WaveSparse &ws, int me, int grp_shift)
{
    const int64_t g0 = fwd ? f.gPos : ix.G2 - (f.gPos + f.gLen);
    const int n_cols = (f.kind == kSimple || f.kind == kPlain || f.kind == kIns) ? f.rLen : (f.kind == kDel ? f.gLen : f.ops_len);
    const uint8_t uni = (f.kind == kSimple || f.kind == kPlain) ? 'M' : (f.kind == kIns ? 'I' : (f.kind == kDel ? 'D' : 0));
    const uint8_t *fo = ops + f.ops_off;
    const uint64_t lt = ((uint64_t)1 << me) - 1, all = WIDTH == 64 ? ~0ull : (((uint64_t)1 << WIDTH) - 1);
    int base_r = 0, base_g = 0;
    for (int c0 = 0; c0 < n_cols; c0 += WIDTH) {
        const int x = c0 + me;
        const bool valid = x < n_cols;
        const uint8_t op = valid ? (uni ? uni : fo[x]) : 0;
        uint64_t mR, mG;
        if (WIDTH == 1) { mR = (op == 'M' || op == 'I'); mG = (op == 'M' || op == 'D'); }
        else { mR = (__ballot(op == 'M' || op == 'I') >> grp_shift) & all; mG = (__ballot(op == 'M' || op == 'D') >> grp_shift) & all; }
        const int ri = base_r + __popcll(mR & lt), gi = base_g + __popcll(mG & lt);
        if (op == 'M') {
            int pl = -1;
            switch (frag_read_char(rd, f, fwd, ri)) { case 'A': pl = kPlA; break; case 'C': pl = kPlC; break; case 'G': pl = kPlG; break; case 'T': pl = kPlT; break; }
            if (pl >= 0) half_inc(pv.pl.h(pl), (uint64_t)(g0 + gi));
        } else if (op == 'I' || op == 'D') {
            const uint8_t prev = x == 0 ? 0 : (uni ? uni : fo[x - 1]);
            if (prev != op) { // first column of a run: this lane records it (:133-150)
                int e = 0;
                while (x + e < n_cols && (uni ? uni : fo[x + e]) == op) e++;
                // a string longer than a record continues in the records behind it ('C'); beyond 255 bases it is refused ('X')
                const int per = (int)sizeof(SparseRec::seq), n_rec = e > 255 ? 1 : (e + per - 1) / per;
                if (e > 255) atomicAdd(sink.refused, 1u);
                const uint32_t at = n_rec > 1 ? atomicAdd(sink.n, (uint32_t)n_rec) : 0u; // (a string and its continuations stay together)
                for (int k = 0; k < n_rec; k++) {
                    SparseRec s; s.pos = g0 + gi - 1; s.type = e > 255 ? 'X' : (k == 0 ? op : 'C');
                    const int lo = k * per, m = e > 255 ? 0 : (e - lo < per ? e - lo : per);
                    for (int i = 0; i < m; i++) s.seq[i] = op == 'I' ? (char)frag_read_char(rd, f, fwd, ri + lo + i) : "ACGT"[ref_code(ix, g0 + gi + lo + i)];
                    s.len = (uint8_t)(k == 0 ? (e > 255 ? 255 : e) : m);
                    if (n_rec == 1) wave_sparse_put(ws, sink, s);
                    else if (at + k < sink.cap) sink.recs[at + k] = s;
                }
            }
        }
        base_r += __popcll(mR); base_g += __popcll(mG);
    }
}

static __device__ __forceinline__ void prof_read(const uint8_t *detail, const DetailLayout &dl, const ReadBatch &rb, const IndexView &ix, const ProfView &pv,
                                                 const SparseSink &sink, const WaveSparse &ws, const uint8_t *admit, int paired, const ColList &cols, uint32_t r)
{
    if (r >= rb.n_reads) return;
    const uint8_t *rec = detail + (uint64_t)r * dl.stride;
    const DetailHdr d = *(const DetailHdr *)rec;
    const Frag *fr = (const Frag *)(rec + sizeof(DetailHdr)) + d.frag0;
    if (d.type == 2) { // UpdateMultiHitCount (:244-271)
        for (int i = 0; i < d.n_frags; i++) { const Frag f = fr[i]; range_add(pv.pl.multi, f.gPos, f.gPos + f.rLen, pv.G); }
        return;
    }
    const uint8_t flag = admit[r];
    if (d.type != 1 || !(flag & 1)) return;
    ReadRef rd;
    const uint32_t off = rb.off[r];
    rd.ascii = rb.bases + off; rd.rlen = (int)(rb.off[r + 1] - off); rd.flipped = (paired && (r & 1)) ? 1 : 0;
    const bool fwd = d.fwd != 0, first = paired ? !(r & 1) : true;
    const bool letters = !(flag & 2); // every base an upper-case ACGT: an exact seed is then a run of "read base == reference base"
    const int strand = first ? (fwd ? kPlF1 : kPlR1) : (fwd ? kPlR2 : kPlF2);
    for (int i = 0; i < d.n_frags; i++) {
        const Frag f = fr[i];
        // every fragment is walked in forward-genome coordinates, like the reference's strings
        // after SelfComplementarySeq: g0 = first genome position under the fragment
        const int64_t g0 = fwd ? f.gPos : ix.G2 - (f.gPos + f.gLen);
        if (i == 0) range_add(pv.pl.h(strand), g0, g0 + rd.rlen, pv.G); // (the reference runs past the array at the genome end)
        if (f.kind == kEmpty) { // an end fragment the quality gate emptied still hits the `gLen == 0` branch (:124/:193) with ""
            SparseRec s; s.pos = g0 - 1; s.type = 'I'; s.len = 0; wave_sparse_put(ws, sink, s);
            continue;
        }
        if (f.kind == kSimple && letters) { range_add(pv.match, g0, g0 + f.rLen, pv.G); continue; }
        bool here = (f.kind == kPlain || f.kind == kSimple) && f.rLen <= 4;
        if (!here) {
            const uint64_t m = __ballot(1); // the lanes that list a fragment now take their places with one atomic
            const int lane = threadIdx.x & 63, leader = __ffsll((unsigned long long)m) - 1;
            uint32_t at = 0;
            if (lane == leader) at = atomicAdd(cols.n, (uint32_t)__popcll(m));
            at = (uint32_t)__shfl((int)at, leader, 64) + (uint32_t)__popcll(m & (((uint64_t)1 << lane) - 1));
            if (at < cols.cap) { ColItem it; it.f = f; it.read = r; it.off = off; it.rlen = rd.rlen; it.flags = (fwd ? 1u : 0u) | (rd.flipped ? 2u : 0u); cols.items[at] = it; }
            else here = true;
        }
        if (here) prof_walk<1>(rd, f, fwd, rec + dl.off_ops, ix, pv, sink, ws, 0, 0);
    }
}

// (the reads in the order of their sorted start keys instead — the lanes of a wavefront then add to neighbouring stretches of the planes —
//  was measured in round 5: 30.38 against 30.46 ms per batch; the kernel is bound by the lines its atomics open, ~24 G line fills and
//  write-backs a second, wherever they lie)
__global__ void __launch_bounds__(256) k_prof_accum(const uint8_t *detail, DetailLayout dl, ReadBatch rb, IndexView ix, ProfView pv,
                                                    SparseSink sink, const uint8_t *admit, int paired, ColList cols)
{
    __shared__ SparseRec s_buf[4][96];
    __shared__ uint32_t s_cnt[4];
    WaveSparse ws; ws.buf = s_buf[threadIdx.x >> 6]; ws.cnt = &s_cnt[threadIdx.x >> 6]; ws.cap = 96;
    if ((threadIdx.x & 63) == 0) *ws.cnt = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    prof_read(detail, dl, rb, ix, pv, sink, ws, admit, paired, cols, r);
    wave_sparse_flush(ws, sink);
}

// pass 3b: the listed fragments, one per group of sixteen lanes
__global__ void __launch_bounds__(256) k_prof_cols(const uint8_t *detail, DetailLayout dl, ReadBatch rb, IndexView ix, ProfView pv,
                                                   SparseSink sink, ColList cols)
{
    __shared__ SparseRec s_buf[4][64];
    __shared__ uint32_t s_cnt[4];
    WaveSparse ws; ws.buf = s_buf[threadIdx.x >> 6]; ws.cnt = &s_cnt[threadIdx.x >> 6]; ws.cap = 64;
    if ((threadIdx.x & 63) == 0) *ws.cnt = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    const uint32_t n = *cols.n < cols.cap ? *cols.n : cols.cap;
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, n_groups = (gridDim.x * blockDim.x) >> 4;
    const int me = threadIdx.x & 15, grp_shift = threadIdx.x & 48;
    const uint32_t first = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 4; // (the wavefront's first group: its lanes leave the loop together)
    for (uint32_t k0 = first; k0 < n; k0 += n_groups) {
        const uint32_t k = k0 + (group - first);
        if (k < n) {
            const ColItem it = cols.items[k];
            ReadRef rd;
            rd.ascii = rb.bases + it.off; rd.rlen = it.rlen; rd.flipped = (it.flags & 2) ? 1 : 0;
            prof_walk<16>(rd, it.f, (it.flags & 1) != 0, detail + (uint64_t)it.read * dl.stride + dl.off_ops, ix, pv, sink, ws, me, grp_shift);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (*ws.cnt >= ws.cap / 2) wave_sparse_flush(ws, sink);
    }
    wave_sparse_flush(ws, sink);
}

// mcx_profile_settle, before the scans: the words of a 16-bit difference plane back to two differences modulo 2^16 (mcx_planes.h)
__global__ void __launch_bounds__(256) k_prof_decode(uint32_t *words, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t w = words[i];
        if (w) words[i] = planes_decode(w);
    }
}

// mcx_profile_settle, after the scans: the exact-seed coverage joins the plane of the reference's base
__global__ void __launch_bounds__(256) k_prof_fold(IndexView ix, ProfView pv)
{
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < pv.G; p += (int64_t)gridDim.x * blockDim.x) {
        const uint16_t m = pv.match[p];
        if (m) pv.pl.h(ref_code(ix, p))[p] += m; // (a 16-bit read-modify-write of the thread's own position)
    }
}

// field widths of MappingRecord_t, applied once all contributions are in (and, on several GPUs,
// after the all-reduce): 12-bit saturation, 16-bit wrap, duplicate cap
__global__ void k_prof_finalize(PlanesView pl, int max_dup)
{
    const int k = blockIdx.y; // one plane per grid row: no division per element (the strand planes are 16-bit words already: nothing to do)
    if (k == kPlMulti) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pl.G; i += (int64_t)gridDim.x * blockDim.x) { const uint32_t v = pl.multi[i]; if (v > 4095u) pl.multi[i] = 4095u; }
        return;
    }
    if (k > kPlReadCount) return;
    const uint16_t top = (uint16_t)(k == kPlReadCount ? max_dup : 4095);
    // two positions per thread: a word of the plane at a time
    uint32_t *w = (uint32_t *)pl.h(k);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pl.stride / 2; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = w[i];
        const uint32_t lo = v & 0xFFFFu, hi = v >> 16;
        if (lo > top || hi > top) w[i] = (lo < top ? lo : top) | ((hi < top ? hi : top) << 16);
    }
}

#endif // __HIPCC__

} // namespace mcx
#endif
