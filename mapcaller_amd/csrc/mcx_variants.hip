// mapcaller_amd/csrc/mcx_variants.hip — variant calling over the alignment profile (the -vcf sink).
//
// Replaces VariantCalling() (reference src/VariantCalling.cpp:696-740), which the reference runs on
// ONE thread (iThreadNum = 1 forced at :717) over a 16-byte record per genome position.
//
// Here the dense part — everything that looks at every position — runs on the GPU over the ten
// planes that mcx_profile_* left in HBM (mcx_planes.h: 22 bytes per position):
//   k_vc_depth   one wavefront per 100-position block: BlockDepthArr             (CalBlockReadDepth :105-121)
//   k_vc_scan    one lane per position: SNV calls, the boundaries of uncovered / duplicated runs,
//                monomorphic records, boundaries of "normal" runs for gVCF       (IdentifyVariants :549-680)
//   k_vc_gather  profile columns of listed positions (for the sparse host logic and the VCF text)
//   k_vc_range   coverage sum / minimum over listed ranges                      (CalRegionCov :197-208, gVCF MIN_DP)
// Both scans are streaming reads (8 B resp. ~13 B per position): HBM-bound.  Whatever the reference
// keeps in std::maps — insert / delete strings, clip break points, discordant pair sites — is sparse
// and stays on the host: indel calls (GetAreaIndFrequency :63-94) are evaluated only at positions that
// have a tally, break points (:173-340) only at candidates, each with columns fetched by k_vc_gather.
// Records the device appends (in any order) are sorted by (position, type) on the host, which is the
// order the reference's single thread produces.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "mcx_variants_host.h"
#include "mcx_planes.h"
#include <hipcub/hipcub.hpp>
#include <mutex>

using namespace mcx;
using namespace mcx_vc;

#define VC_TRY(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return mcx_set_error(MCX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {

__global__ void __launch_bounds__(256) k_vc_depth(PlanesView pl, int64_t G, int64_t n_blocks, int32_t *depth)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= n_blocks) return;
    uint32_t sum = 0;
    for (int o = lane; o < kBlock; o += 64) {
        const int64_t g = b * kBlock + o;
        if (g < G) sum += pl.get(kPlA, g) + pl.get(kPlC, g) + pl.get(kPlG, g) + pl.get(kPlT, g);
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
    if (lane == 0) depth[b] = sum > 0 ? (int32_t)(sum / kBlock) : 0;
}

// One lane per position; a workgroup walks a contiguous range of 256-position tiles, so the left
// neighbour's class is carried from tile to tile (only a range's first tile evaluates it again).
// Records are staged in LDS and flushed with ONE global atomic per ~1000 records: a counter bumped
// per wavefront serialises on its L2 atomic unit (measured: 9 ns per bump, 10x the streaming time).
enum { kScanTile = 256, kScanStage = 1024 }; // (a tile appends at most 4 x 256 records)

__global__ void __launch_bounds__(kScanTile) k_vc_scan(PlanesView pl, const int32_t *depth, IndexView ix, ScanParams sp, SiteRec *out,
                                                       unsigned long long *n_out, uint64_t cap, int64_t tiles_per_group)
{
    __shared__ SiteRec stage[kScanStage];
    __shared__ uint32_t wave_sum[kScanTile / 64];
    __shared__ uint32_t n_staged, carry; // carry: class | cand << 2 of the position left of the tile
    __shared__ unsigned long long flush_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n_tiles = (sp.G + kScanTile - 1) / kScanTile;
    const int64_t t0 = (int64_t)blockIdx.x * tiles_per_group, t1 = min(n_tiles, t0 + tiles_per_group);
    if (t0 >= t1) return;
    if (tid == 0) {
        n_staged = 0; carry = 0;
        if (t0 > 0) { const SiteEval p = eval_site(pl, depth, ix, sp, t0 * kScanTile - 1); carry = (uint32_t)p.cls | ((uint32_t)p.cand << 2); }
    }
    __syncthreads();
    auto flush = [&]() { // all threads; n_staged is stable on entry
        const uint32_t n = n_staged;
        if (tid == 0 && n) flush_base = atomicAdd(n_out, (unsigned long long)n);
        __syncthreads();
        for (uint32_t i = tid; i < n; i += kScanTile) { const uint64_t at = flush_base + i; if (at < cap) out[at] = stage[i]; }
        __syncthreads();
        if (tid == 0) n_staged = 0;
        __syncthreads();
    };
    for (int64_t t = t0; t < t1; t++) {
        const int64_t g = t * kScanTile + tid;
        const bool live = g < sp.G;
        SiteEval e;
        e.cls = 0; e.cand = false; e.call = false;
        if (live) e = eval_site(pl, depth, ix, sp, g);
        uint32_t mine = (uint32_t)e.cls | ((uint32_t)e.cand << 2);
        uint32_t left = __shfl_up(mine, 1, 64);
        __shared__ uint32_t wave_last[kScanTile / 64];
        if (lane == 63) wave_last[wave] = mine;
        __syncthreads();
        if (lane == 0) left = wave == 0 ? carry : wave_last[wave - 1];
        const int prev_cls = (int)(left & 3);
        const bool prev_cand = (left >> 2) != 0;
        // up to four records per position, in this order
        const bool r_end = live && prev_cls != 0 && e.cls != prev_cls;
        const bool r_start = live && e.cls != 0 && e.cls != prev_cls;
        const bool r_site = live && (e.call || (e.cand && sp.mono));
        const bool r_norm = live && sp.gvcf && e.cand != prev_cand;
        const uint32_t n = (uint32_t)r_end + r_start + r_site + r_norm;
        uint32_t incl = n; // inclusive scan within the wave
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int w = 0; w < kScanTile / 64; w++) { if (w < wave) before += wave_sum[w]; total += wave_sum[w]; }
        if (n_staged + total > kScanStage) flush(); // uniform across the group
        if (n) {
            uint32_t at = n_staged + before + incl - n;
            SiteRec ev = e.rec; ev.geno = ev.qscore = 0; ev.alt = 0xFF; ev.DP = ev.AD_ref = ev.AD_alt = 0;
            if (r_end) { ev.type = prev_cls == 1 ? eGapEnd : eDupEnd; stage[at++] = ev; }
            if (r_start) { ev.type = e.cls == 1 ? eGapStart : eDupStart; stage[at++] = ev; }
            if (r_site) stage[at++] = e.rec;
            if (r_norm) { ev.type = e.cand ? eNormStart : eNormEnd; stage[at++] = ev; }
        }
        __syncthreads();
        if (tid == kScanTile - 1) { n_staged += total; carry = mine; }
        __syncthreads();
    }
    flush();
}

// (position, type) order of the appended records: sort keys, then move the records
__global__ void k_vc_keys(const SiteRec *recs, uint64_t n, uint64_t *keys, uint32_t *idx)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = ((uint64_t)recs[i].pos << 8) | recs[i].type;
    idx[i] = (uint32_t)i;
}

__global__ void k_vc_permute(const SiteRec *recs, const uint32_t *idx, uint64_t n, SiteRec *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = recs[idx[i]];
}

__global__ void k_vc_gather(PlanesView pl, const int32_t *depth, IndexView ix, int64_t G, const int64_t *pos, uint64_t n, Column *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t g = pos[i];
    Column c;
    if (g < 0 || g >= G) { memset(&c, 0, sizeof c); out[i] = c; return; }
    for (int k = 0; k < nPlanes; k++) c.v[k] = pl.get(k, g);
    c.depth = depth[g / kBlock];
    c.ref = (uint32_t)ref_code(ix, g);
    out[i] = c;
}

__global__ void __launch_bounds__(256) k_vc_range(PlanesView pl, int64_t G, const RangeQ *q, uint64_t n, unsigned long long *out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    const RangeQ rq = q[i];
    unsigned long long acc = rq.mode ? ~0ull : 0ull;
    for (int64_t g = rq.beg + lane; g <= rq.end; g += 64) {
        if (g < 0 || g >= G) continue;
        const unsigned long long cov = (unsigned long long)pl.get(kPlA, g) + pl.get(kPlC, g) + pl.get(kPlG, g) + pl.get(kPlT, g);
        if (rq.mode) { if (cov > 0 && cov < acc) acc = cov; }
        else acc += cov;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = (unsigned long long)__shfl_down((long long)acc, o, 64);
        acc = rq.mode ? (other < acc ? other : acc) : acc + other;
    }
    if (lane == 0) out[i] = acc;
}

// Pageable host memory <-> device through the two halves of a page-locked buffer (a direct copy runs at 1-2 GB/s; the lists
// here are hundreds of MB): one half is on the bus while the other is copied by the host.
struct Bounce {
    static constexpr size_t kHalf = (size_t)16 << 20;
    std::mutex mu;
    uint8_t *pin = nullptr;
    hipStream_t stream = nullptr;
    int ready()
    {
        if (pin) return 0;
        VC_TRY(hipHostMalloc((void **)&pin, 2 * kHalf));
        VC_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        return 0;
    }
};
static Bounce g_bounce;

static int copy_out(void *dst, const void *d_src, size_t bytes)
{
    Bounce &b = g_bounce;
    std::lock_guard<std::mutex> lock(b.mu);
    if (int rc = b.ready()) return rc;
    VC_TRY(hipDeviceSynchronize()); // (the producers ran on the null stream)
    if (bytes >= ((size_t)8 << 20)) { // fresh pages: touched by many threads now instead of by the one copy below, fault by fault
        const size_t pages = (bytes + 4095) / 4096;
        volatile uint8_t *d = (volatile uint8_t *)dst;
        par_ranges(pages, vc_threads(pages, 1024), [&](unsigned, size_t lo, size_t hi) { for (size_t p = lo; p < hi; p++) d[p * 4096] = 0; });
    }
    const size_t n_piece = (bytes + Bounce::kHalf - 1) / Bounce::kHalf;
    auto start = [&](size_t k) { return hipMemcpyAsync(b.pin + (k & 1) * Bounce::kHalf, (const uint8_t *)d_src + k * Bounce::kHalf, std::min(Bounce::kHalf, bytes - k * Bounce::kHalf), hipMemcpyDeviceToHost, b.stream); };
    if (n_piece) VC_TRY(start(0));
    for (size_t k = 0; k < n_piece; k++) {
        VC_TRY(hipStreamSynchronize(b.stream));
        if (k + 1 < n_piece) VC_TRY(start(k + 1));
        memcpy((uint8_t *)dst + k * Bounce::kHalf, b.pin + (k & 1) * Bounce::kHalf, std::min(Bounce::kHalf, bytes - k * Bounce::kHalf));
    }
    return 0;
}

static int copy_in(void *d_dst, const void *src, size_t bytes)
{
    Bounce &b = g_bounce;
    std::lock_guard<std::mutex> lock(b.mu);
    if (int rc = b.ready()) return rc;
    const size_t n_piece = (bytes + Bounce::kHalf - 1) / Bounce::kHalf;
    for (size_t k = 0; k < n_piece; k++) {
        const size_t m = std::min(Bounce::kHalf, bytes - k * Bounce::kHalf);
        memcpy(b.pin + (k & 1) * Bounce::kHalf, (const uint8_t *)src + k * Bounce::kHalf, m); // (the copy of piece k - 1 is on the bus meanwhile)
        if (k >= 1) VC_TRY(hipStreamSynchronize(b.stream)); // piece k - 1 has left its half; piece k - 2's half was free already
        VC_TRY(hipMemcpyAsync((uint8_t *)d_dst + k * Bounce::kHalf, b.pin + (k & 1) * Bounce::kHalf, m, hipMemcpyHostToDevice, b.stream));
    }
    VC_TRY(hipStreamSynchronize(b.stream));
    return 0;
}

template <typename T> struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { if (p) { (void)hipFree(p); p = nullptr; } VC_TRY(hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T))); return 0; }
};

// the dense half on the GPU: the ten planes in HBM behind the caller's interface
class GpuProfile : public DenseProfile {
public:
    GpuProfile(const mcx_index *ix, const uint32_t *planes) : ix_(ix), pl_(planes_view((void *)planes, ix->view.G)), G_(ix->view.G) {}
    int64_t genome_size() const override { return G_; }
    int gather(const std::vector<int64_t> &pos, ColVec &out) override
    {
        out.resize(pos.size());
        if (pos.empty()) return 0;
        DevBuf<int64_t> d_pos; DevBuf<Column> d_col;
        int rc;
        if ((rc = d_pos.alloc(pos.size())) || (rc = d_col.alloc(pos.size()))) return rc;
        if ((rc = copy_in(d_pos.p, pos.data(), pos.size() * sizeof(int64_t)))) return rc;
        k_vc_gather<<<(unsigned)((pos.size() + 255) / 256), 256>>>(pl_, d_depth_.p, ix_->view, G_, d_pos.p, pos.size(), d_col.p);
        VC_TRY(hipGetLastError());
        return copy_out(out.data(), d_col.p, pos.size() * sizeof(Column));
    }

    int ranges(const std::vector<RangeQ> &q, std::vector<unsigned long long> &out) override
    {
        out.resize(q.size());
        if (q.empty()) return 0;
        DevBuf<RangeQ> d_q; DevBuf<unsigned long long> d_out;
        int rc;
        if ((rc = d_q.alloc(q.size())) || (rc = d_out.alloc(q.size()))) return rc;
        VC_TRY(hipMemcpy(d_q.p, q.data(), q.size() * sizeof(RangeQ), hipMemcpyHostToDevice));
        k_vc_range<<<(unsigned)((q.size() + 3) / 4), 256>>>(pl_, G_, d_q.p, q.size(), d_out.p);
        VC_TRY(hipGetLastError());
        VC_TRY(hipMemcpy(out.data(), d_out.p, q.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        return 0;
    }

    // block depth, then the per-position scan; the record list is sized by a first guess and, if the
    // scan produced more, by its own count
    int scan(const ScanParams &sp_in, SiteVec &sites, double &ms_depth_, double &ms_scan_) override
    {
        SubLap lap;
        const int64_t nb = (G_ + kBlock - 1) / kBlock;
        int rc;
        if ((rc = d_depth_.alloc((size_t)nb))) return rc;
        lap("scan: depth array");
        hipEvent_t ev[4];
        for (auto &e : ev) VC_TRY(hipEventCreate(&e));
        VC_TRY(hipEventRecord(ev[0], 0));
        k_vc_depth<<<(unsigned)((nb + 3) / 4), 256>>>(pl_, G_, nb, d_depth_.p);
        VC_TRY(hipGetLastError());
        VC_TRY(hipEventRecord(ev[1], 0));
        const ScanParams sp = sp_in;
        DevBuf<SiteRec> d_out; DevBuf<unsigned long long> d_n;
        if ((rc = d_n.alloc(1))) return rc;
        uint64_t cap = std::max<uint64_t>(1u << 20, (uint64_t)G_ / 64); // (a first guess: a list that runs over is sized by its own count and the scan repeated)
        unsigned long long n = 0;
        for (int attempt = 0; attempt < 2; attempt++) {
            lap("scan: depth kernel queued");
            if ((rc = d_out.alloc(cap))) return rc;
            lap("scan: record list");
            VC_TRY(hipMemset(d_n.p, 0, sizeof(unsigned long long)));
            VC_TRY(hipEventRecord(ev[2], 0));
            const int64_t n_tiles = (G_ + kScanTile - 1) / kScanTile, groups = std::min<int64_t>(n_tiles, 256 * 32);
            const int64_t per_group = (n_tiles + groups - 1) / groups;
            k_vc_scan<<<(unsigned)((n_tiles + per_group - 1) / per_group), kScanTile>>>(pl_, d_depth_.p, ix_->view, sp, d_out.p, d_n.p, cap, per_group);
            VC_TRY(hipGetLastError());
            VC_TRY(hipEventRecord(ev[3], 0));
            VC_TRY(hipMemcpy(&n, d_n.p, sizeof n, hipMemcpyDeviceToHost));
            if (n <= cap) break;
            if (attempt == 1) return mcx_set_error(MCX_ERR_CAPACITY, "variant scan: record list overflow");
            cap = n;
        }
        float ms = 0;
        VC_TRY(hipEventElapsedTime(&ms, ev[0], ev[1])); ms_depth_ = ms;
        VC_TRY(hipEventElapsedTime(&ms, ev[2], ev[3])); ms_scan_ = ms;
        for (auto &e : ev) (void)hipEventDestroy(e);
        lap("scan: kernels");
        sites.resize(n);
        if (n == 0) return 0;
        if (n >= (1ull << 32)) return mcx_set_error(MCX_ERR_CAPACITY, "variant scan: more than 2^32 records");
        // bring the appended records into (position, type) order on the device
        // (one allocation for the keys, the record numbers, the sort's scratch and the ordered records: a device allocation
        //  and its release cost milliseconds each)
        size_t tmp_bytes = 0;
        {
            hipcub::DoubleBuffer<uint64_t> dk0((uint64_t *)nullptr, (uint64_t *)nullptr);
            hipcub::DoubleBuffer<uint32_t> dv0((uint32_t *)nullptr, (uint32_t *)nullptr);
            VC_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, dk0, dv0, (int64_t)n, 0, 48));
        }
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t o_k0 = 0, o_k1 = o_k0 + up(n * 8), o_i0 = o_k1 + up(n * 8), o_i1 = o_i0 + up(n * 4), o_sorted = o_i1 + up(n * 4),
                     o_tmp = o_sorted + up(n * sizeof(SiteRec)), arena_bytes = o_tmp + up(tmp_bytes);
        DevBuf<uint8_t> arena;
        if ((rc = arena.alloc(arena_bytes))) return rc;
        lap("scan: room for the sort");
        uint64_t *k0 = (uint64_t *)(arena.p + o_k0), *k1 = (uint64_t *)(arena.p + o_k1);
        uint32_t *i0 = (uint32_t *)(arena.p + o_i0), *i1 = (uint32_t *)(arena.p + o_i1);
        SiteRec *d_sorted = (SiteRec *)(arena.p + o_sorted);
        const unsigned nb256 = (unsigned)((n + 255) / 256);
        k_vc_keys<<<nb256, 256>>>(d_out.p, n, k0, i0);
        hipcub::DoubleBuffer<uint64_t> dk(k0, k1);
        hipcub::DoubleBuffer<uint32_t> dv(i0, i1);
        VC_TRY(hipcub::DeviceRadixSort::SortPairs(arena.p + o_tmp, tmp_bytes, dk, dv, (int64_t)n, 0, 48)); // 40 position bits + 8 type bits
        if (lap.on) { VC_TRY(hipDeviceSynchronize()); lap("scan: keys + radix sort"); }
        k_vc_permute<<<nb256, 256>>>(d_out.p, dv.Current(), n, d_sorted);
        VC_TRY(hipGetLastError());
        VC_TRY(hipDeviceSynchronize());
        lap("scan: order on the device");
        return copy_out(sites.data(), d_sorted, n * sizeof(SiteRec));
    }

private:
    const mcx_index *ix_;
    PlanesView pl_;
    int64_t G_;
    DevBuf<int32_t> d_depth_;
};

} // namespace

extern "C" void mcx_vcf_defaults(mcx_vcf_opts *o)
{
    memset(o, 0, sizeof *o);
    o->ploidy = 2; o->min_allele_depth = 5; o->min_cnv = 50; o->min_gap = 50; o->fragment_size = 500; // main.cpp:157-187
    o->max_dup = 5; o->max_clip = 5; o->freq_thr = 0.2f; o->sample_id = "unknown";
}

extern "C" int mcx_planes_alloc(const mcx_index *ix, uint32_t **d_planes)
{
    if (!ix || !d_planes) return mcx_set_error(MCX_ERR_ARG, "mcx_planes_alloc: null argument");
    VC_TRY(hipSetDevice(ix->device));
    const size_t bytes = (size_t)planes_bytes(ix->view.G);
    VC_TRY(hipMalloc((void **)d_planes, bytes));
    VC_TRY(hipMemset(*d_planes, 0, bytes));
    return 0;
}

extern "C" void mcx_planes_free(uint32_t *d_planes) { if (d_planes) (void)hipFree(d_planes); }
extern "C" uint64_t mcx_planes_bytes(int64_t genome_size) { return planes_bytes(genome_size); }

extern "C" int mcx_call_variants(const mcx_index *ix, const uint32_t *d_planes, const mcx_sparse_rec *recs, uint64_t n_recs,
                                 int64_t paired_pairs, int64_t pair_dist_sum, int64_t pair_len_sum,
                                 const mcx_vcf_opts *opts, const char *vcf_path, mcx_vcf_stats *stats)
{
    if (!ix || !d_planes || !opts || !vcf_path || (n_recs && !recs)) return mcx_set_error(MCX_ERR_ARG, "mcx_call_variants: null argument");
    if (opts->min_allele_depth < 1) return mcx_set_error(MCX_ERR_UNSUPPORTED, "mcx_call_variants: -ad must be at least 1");
    VC_TRY(hipSetDevice(ix->device));
    const auto t0 = std::chrono::steady_clock::now();
    mcx_vcf_opts o = *opts;
    if (o.gvcf && o.monomorphic) o.gvcf = 0; // main.cpp:322
    GpuProfile prof(ix, d_planes);
    Caller c(ix->host, ix->view.G2, prof, o);
    const int rc = c.run(recs, n_recs, paired_pairs, pair_dist_sum, pair_len_sum, vcf_path, stats);
    if (rc == 0 && stats) stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}
