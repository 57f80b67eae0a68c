// mapcaller_amd/csrc/mcx_profile.h — the -vcf bookkeeping of the mapping loop on the GPU (device only).
//
// Replaces UpdateProfile / UpdateMultiHitCount (reference src/AlignmentProfile.cpp:41-271), which
// the reference runs serially under ProfileLock (src/ReadMapping.cpp:562-573) into a 16-byte
// bit-field record per genome position (MappingRecord_t, src/structure.h:152-163).
//
// Here the per-position counters are ten u32 planes [k][G] (A C G T multi_hit readCount F1 R2 F2 R1)
// in caller-owned HBM, so that a multi-GPU run can sum them with one RCCL all-reduce; the field
// widths of the reference (12-bit saturation at 4095, 16-bit wrap, duplicate cap) are applied
// afterwards by k_prof_finalize.  One wavefront accumulates one read: lanes are consecutive
// alignment columns, so each atomic instruction covers a contiguous run of a plane.  The only
// order-dependent rule — at most iMaxDuplicate uniquely mapped reads are admitted per start
// position, in input order (AlignmentProfile.cpp:76-77) — is decided before accumulation by
// sorting (start, read index) keys.  Insert / delete strings and break points are sparse
// records appended to a list that the host folds into maps.
#ifndef MCX_PROFILE_H
#define MCX_PROFILE_H
#include "mcx_glue.h"

namespace mcx {

#if defined(__HIPCC__)

enum { kPlA = 0, kPlC, kPlG, kPlT, kPlMulti, kPlReadCount, kPlF1, kPlR2, kPlF2, kPlR1, kPlanes };

struct ProfView {
    uint32_t *plane;  // [kPlanes][G]
    int64_t G;
    int32_t max_dup, max_clip;
};

struct SparseSink { SparseRec *recs; uint32_t *n; uint32_t cap; uint32_t *refused; };

static __device__ __forceinline__ void sparse_put(const SparseSink &s, const SparseRec &r)
{
    const uint32_t at = atomicAdd(s.n, 1u);
    if (at < s.cap) s.recs[at] = r;
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
            const Frag *f = (const Frag *)(rec + sizeof(DetailHdr));
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
__global__ void k_prof_admit(const uint64_t *keys, uint64_t n, ProfView pv, uint8_t *admit, uint32_t own_lo, uint32_t n_own)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t key = keys[j];
    const uint32_t idx = (uint32_t)key - own_lo;
    if (idx >= n_own) return;
    const uint64_t g = key >> 32;
    int rank = 0;
    for (int k = 1; k <= pv.max_dup && (uint64_t)k <= j; k++) { if ((keys[j - k] >> 32) == g) rank++; else break; }
    const uint32_t before = pv.plane[(uint64_t)kPlReadCount * pv.G + g];
    admit[idx] = (before + (uint32_t)rank < (uint32_t)pv.max_dup) ? 1 : 0;
}

// pass 2b (after every flag is out): the first key of each start position adds the round's admissions to
// readCount — the count of the whole run, the same on every shard (`readCount < iMaxDuplicate` then ++, :76-77)
__global__ void k_prof_count(const uint64_t *keys, uint64_t n, ProfView pv)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t g = keys[j] >> 32;
    if (j > 0 && (keys[j - 1] >> 32) == g) return;
    uint32_t run = 1;
    while (run < (uint32_t)pv.max_dup && j + run < n && (keys[j + run] >> 32) == g) run++;
    uint32_t *cnt = &pv.plane[(uint64_t)kPlReadCount * pv.G + g];
    const uint32_t v = *cnt + run;
    *cnt = v < (uint32_t)pv.max_dup ? v : (uint32_t)pv.max_dup;
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

// pass 3: one wavefront per read
__global__ void __launch_bounds__(256) k_prof_accum(const uint8_t *detail, DetailLayout dl, ReadBatch rb, IndexView ix, ProfView pv,
                                                    SparseSink sink, const uint8_t *admit, int paired)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (uint32_t r = wave; r < rb.n_reads; r += n_waves) {
        const uint8_t *rec = detail + (uint64_t)r * dl.stride;
        const DetailHdr &d = *(const DetailHdr *)rec;
        const Frag *fr = (const Frag *)(rec + sizeof(DetailHdr));
        const uint8_t *ops = rec + dl.off_ops;
        if (d.type == 2) { // UpdateMultiHitCount (:244-271)
            for (int i = 0; i < d.n_frags; i++) {
                const int64_t g0 = fr[i].gPos;
                for (int o = lane; o < fr[i].rLen; o += 64)
                    if (g0 + o >= 0 && g0 + o < pv.G) atomicAdd(&pv.plane[(uint64_t)kPlMulti * pv.G + g0 + o], 1u);
            }
            continue;
        }
        if (d.type != 1 || !admit[r]) continue;
        ReadRef rd;
        rd.ascii = rb.bases + rb.off[r]; rd.rlen = (int)(rb.off[r + 1] - rb.off[r]); rd.flipped = (paired && (r & 1)) ? 1 : 0;
        const bool fwd = d.fwd != 0, first = paired ? !(r & 1) : true;
        const Frag &a = fr[0];
        const int64_t start = fwd ? a.gPos : ix.G2 - (a.gPos + a.gLen);
        const int strand = first ? (fwd ? kPlF1 : kPlR1) : (fwd ? kPlR2 : kPlF2);
        for (int o = lane; o < rd.rlen; o += 64)
            if (start + o < pv.G) atomicAdd(&pv.plane[(uint64_t)strand * pv.G + start + o], 1u); // (the reference runs past the array at the genome end)
        for (int i = 0; i < d.n_frags; i++) {
            const Frag &f = fr[i];
            // every fragment is walked in forward-genome coordinates, like the reference's strings
            // after SelfComplementarySeq: g0 = first genome position under the fragment
            const int64_t g0 = fwd ? f.gPos : ix.G2 - (f.gPos + f.gLen);
            if (f.kind == kEmpty) { // an end fragment the quality gate emptied still hits the `gLen == 0` branch (:124/:193) with ""
                if (lane == 0) { SparseRec s; s.pos = g0 - 1; s.type = 'I'; s.len = 0; sparse_put(sink, s); }
                continue;
            }
            const int n_cols = (f.kind == kSimple || f.kind == kPlain || f.kind == kIns) ? f.rLen : (f.kind == kDel ? f.gLen : f.ops_len);
            const uint8_t uni = (f.kind == kSimple || f.kind == kPlain) ? 'M' : (f.kind == kIns ? 'I' : (f.kind == kDel ? 'D' : 0));
            int base_r = 0, base_g = 0;
            for (int c0 = 0; c0 < n_cols; c0 += 64) {
                const int x = c0 + lane;
                const bool valid = x < n_cols;
                const uint8_t op = valid ? (uni ? uni : ops[f.ops_off + x]) : 0;
                const uint64_t mR = __ballot(op == 'M' || op == 'I'), mG = __ballot(op == 'M' || op == 'D');
                const int ri = base_r + __popcll(mR & lt_mask), gi = base_g + __popcll(mG & lt_mask);
                if (op == 'M') {
                    int pl = -1;
                    switch (frag_read_char(rd, f, fwd, ri)) { case 'A': pl = kPlA; break; case 'C': pl = kPlC; break; case 'G': pl = kPlG; break; case 'T': pl = kPlT; break; }
                    if (pl >= 0) atomicAdd(&pv.plane[(uint64_t)pl * pv.G + g0 + gi], 1u);
                } else if (op == 'I' || op == 'D') {
                    const uint8_t prev = x == 0 ? 0 : (uni ? uni : ops[f.ops_off + x - 1]);
                    if (prev != op) { // first column of a run: this lane records it (:133-150)
                        int e = 0;
                        while (x + e < n_cols && (uni ? uni : ops[f.ops_off + x + e]) == op) e++;
                        // a string longer than a record continues in the records behind it ('C'); beyond 255 bases it is refused ('X')
                        const int per = (int)sizeof(SparseRec::seq), n_rec = e > 255 ? 1 : (e + per - 1) / per;
                        const uint32_t at = atomicAdd(sink.n, (uint32_t)n_rec);
                        if (e > 255) atomicAdd(sink.refused, 1u);
                        for (int k = 0; k < n_rec; k++) {
                            SparseRec s; s.pos = g0 + gi - 1; s.type = e > 255 ? 'X' : (k == 0 ? op : 'C');
                            const int lo = k * per, m = e > 255 ? 0 : (e - lo < per ? e - lo : per);
                            for (int i = 0; i < m; i++) s.seq[i] = op == 'I' ? (char)frag_read_char(rd, f, fwd, ri + lo + i) : "ACGT"[ref_code(ix, g0 + gi + lo + i)];
                            s.len = (uint8_t)(k == 0 ? (e > 255 ? 255 : e) : m);
                            if (at + k < sink.cap) sink.recs[at + k] = s;
                        }
                    }
                }
                base_r += __popcll(mR); base_g += __popcll(mG);
            }
        }
    }
}

// field widths of MappingRecord_t, applied once all contributions are in (and, on several GPUs,
// after the all-reduce): 12-bit saturation, 16-bit wrap, duplicate cap
__global__ void k_prof_finalize(uint32_t *plane, int64_t G, int max_dup)
{
    const int k = blockIdx.y; // one plane per grid row: no division per element
    uint32_t *p = plane + (uint64_t)k * G;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < G; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t v = p[i];
        if (k <= kPlMulti) v = v < 4095u ? v : 4095u;
        else if (k == kPlReadCount) v = v < (uint32_t)max_dup ? v : (uint32_t)max_dup;
        else v &= 0xFFFFu;
        p[i] = v;
    }
}

#endif // __HIPCC__

} // namespace mcx
#endif
