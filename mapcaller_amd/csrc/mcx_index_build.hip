// mapcaller_amd/csrc/mcx_index_build.hip — BWA-compatible FM-index construction on the GPU.
//
// Replaces bwa_idx_build (reference src/BWT_Index/bwtindex.c:77-160): bns_fasta2bntseq
// (bntseq.c:158-230) packs the FASTA, bwt_bwtgen2 builds the BWT of forward+reverse-complement
// text on the CPU (bwt_gen.c, ~1 h for a human genome), bwt_bwtupdate_core (:53-75) interleaves
// the occurrence counts and bwt_cal_sa (bwt.c:101-125) samples the suffix array every 32 rows.
// The output files are byte-identical; the construction is not a port:
//
//   text T = X . revcomp(X) stays in HBM as one byte per base; the suffix array is built by
//   prefix doubling with hipCUB radix sorts (first pass on 16-base packed keys, then rank
//   pairs (r[i], r[i+h]) with h = 16, 32, ...), bucket by bucket of the first bases so that keys
//   stay within 64 bits and sort buffers at bucket size; BWT / occ blocks / SA samples are derived
//   by one pass over the finished array.  Sized for one GPU's HBM (about 28 bytes per text
//   position: a GRCh38-sized genome, 6.2 G positions, peaks near 175 GB); texts of 2^33 positions
//   or more are refused (MCX_ERR_UNSUPPORTED).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>

#include "../../include/mcx.h"
#include "mcx_host.h"
#include "mcx_types.h"
#include "mcx_build.h"
#include "mcx_fm.h"

using namespace mcx;


#define HIP_TRYB(expr)                                                                            \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return mcx_set_error(MCX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// T[i] = X[i] (i < G), T[G + i] = 3 - X[G - 1 - i]
__global__ void k_make_text(const uint8_t *fwd, uint64_t G, uint8_t *T)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < G; i += (uint64_t)gridDim.x * blockDim.x) {
        uint8_t c = fwd[i];
        T[i] = c;
        T[2 * G - 1 - i] = (uint8_t)(3 - c);
    }
}

// ---- suffix sorting: prefix doubling over buckets -------------------------------------------------
// Suffix i lives in bucket b(i) = its first kb bases (positions past the end read as A): buckets are
// in suffix order, so a suffix's rank is its bucket's base + its rank inside the bucket.  Every
// bucket holds fewer than 2^31 suffixes, which keeps a doubling key — (rank inside the bucket,
// global rank of the suffix h further on) — within 64 bits for texts up to 2^33 positions, and the
// sort buffers at bucket size.  Small texts are one bucket.
struct Buckets {
    int kb, n;                 // bases per bucket id, number of buckets (4^kb, at most 256)
    uint64_t base[257];        // first SA slot of each bucket; base[n] = N
};

static __device__ __forceinline__ uint32_t bucket_of(const uint8_t *T, uint64_t N, uint64_t i, int kb)
{
    uint32_t b = 0;
    for (int j = 0; j < kb; j++) b = (b << 2) | (i + j < N ? T[i + j] : 0u);
    return b;
}

__global__ void k_bucket_count(const uint8_t *T, uint64_t N, int kb, unsigned long long *count)
{
    __shared__ unsigned int local[256];
    for (int k = threadIdx.x; k < 256; k += blockDim.x) local[k] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (uint64_t)gridDim.x * blockDim.x) atomicAdd(&local[bucket_of(T, N, i, kb)], 1u);
    __syncthreads();
    for (int k = threadIdx.x; k < 256; k += blockDim.x) if (local[k]) atomicAdd(&count[k], (unsigned long long)local[k]);
}

// positions into their buckets' SA segments (any order inside a bucket: it is sorted next); one
// tile of 4096 positions per step, one global reservation per bucket and tile
__global__ void __launch_bounds__(256) k_bucket_scatter(const uint8_t *T, uint64_t N, int kb, unsigned long long *cursor, uint64_t *sa)
{
    __shared__ unsigned int cnt[256];
    __shared__ unsigned long long at[256];
    const uint64_t tile = 4096, n_tiles = (N + tile - 1) / tile;
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        for (int k = threadIdx.x; k < 256; k += blockDim.x) cnt[k] = 0;
        __syncthreads();
        uint32_t mine[16], slot[16];
        for (int s = 0; s < 16; s++) {
            const uint64_t i = t * tile + (uint64_t)s * 256 + threadIdx.x;
            mine[s] = 0xFFFFFFFFu;
            if (i < N) { mine[s] = bucket_of(T, N, i, kb); slot[s] = atomicAdd(&cnt[mine[s]], 1u); }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < 256; k += blockDim.x) if (cnt[k]) at[k] = atomicAdd(&cursor[k], (unsigned long long)cnt[k]);
        __syncthreads();
        for (int s = 0; s < 16; s++) {
            const uint64_t i = t * tile + (uint64_t)s * 256 + threadIdx.x;
            if (mine[s] != 0xFFFFFFFFu) sa[at[mine[s]] + slot[s]] = i;
        }
        __syncthreads();
    }
}

// first-pass key of the suffix at sa[j]: 16 bases packed MSB first (positions past the end read as
// 0) and, below them, the number of real bases (a suffix that runs into the terminator sorts before
// a longer one with the same padded prefix)
__global__ void k_init_keys(const uint8_t *T, uint64_t N, const uint64_t *sa, uint64_t n, uint64_t *key)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = sa[j];
        uint64_t k = 0;
        const uint64_t rem = N - i;
        const int real = rem < 16 ? (int)rem : 16;
        for (int c = 0; c < 16; c++) k = (k << 2) | (c < real ? T[i + c] : 0);
        key[j] = (k << 5) | (uint64_t)real;
    }
}

// head[j] = j if the sorted key at j differs from its predecessor, else 0 (j = 0 is a head); j counts inside the bucket
__global__ void k_mark_heads(const uint64_t *key, uint64_t n, uint32_t *head)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x)
        head[j] = (j == 0 || key[j] != key[j - 1]) ? (uint32_t)j : 0u;
}

// after an inclusive max-scan head[j] is the first index of j's group: rank = bucket base + that + 1
__global__ void k_scatter_rank(const uint64_t *sa, const uint32_t *head, uint64_t n, uint64_t base, uint64_t *rank, unsigned long long *n_groups)
{
    unsigned long long local = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        rank[sa[j]] = base + head[j] + 1;
        if (head[j] == (uint32_t)j) local++;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(n_groups, local);
}

// doubling key for the suffix at sa[j] of a bucket: (rank inside the bucket, rank[i + h]) with 0 past the end
__global__ void k_pair_keys(const uint64_t *sa, const uint64_t *rank, uint64_t n, uint64_t base, uint64_t N, uint64_t h, uint64_t *key)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = sa[j];
        const uint64_t r2 = i + h < N ? rank[i + h] : 0;
        key[j] = ((rank[i] - base) << 33) | r2;
    }
}

struct BuildOut {
    uint32_t *bwt;      // occ-interleaved words
    uint64_t *sa;       // sampled, sa[0] = ~0
    uint64_t *sa_full;  // optional, N + 1 entries
    unsigned long long *primary;
};

// Row r of the sorted matrix of T$ (r in [0, N]): row 0 is "$" (SA = N), row r >= 1 is idx[r-1].
// BWT char of a row = T[SA - 1]; the row with SA = 0 is `primary` and is skipped in the packed
// string, so string position m = r - (r > primary).  One thread packs 16 symbols (one word).
__global__ void k_pack_bwt(const uint8_t *T, const uint64_t *idx, uint64_t N, uint64_t primary, uint32_t *words)
{
    const uint64_t n_words = (N + 15) / 16;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = 0;
        for (int s = 0; s < 16; s++) {
            uint64_t m = w * 16 + s;
            uint32_t c = 0;
            if (m < N) {
                uint64_t r = m + (m >= primary);
                uint64_t sa = r == 0 ? N : idx[r - 1];
                c = T[sa - 1];
            }
            v = (v << 2) | c;
        }
        words[w] = v;
    }
}

// per 128-symbol block: counts of A,C,G,T inside the block (to be prefix-summed)
__global__ void k_block_counts(const uint32_t *words, uint64_t N, uint64_t n_blocks, uint64_t *cnt /* 4 x n_blocks, base-major */)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t c[4] = {0, 0, 0, 0};
        for (int j = 0; j < 8; j++) {
            uint64_t w = b * 8 + j;
            uint64_t first = w * 16;
            if (first >= N) break;
            int m = N - first < 16 ? (int)(N - first) : 16;
            uint32_t v = words[w];
            for (int s = 0; s < m; s++) c[(v >> (30 - 2 * s)) & 3]++;
        }
        for (int k = 0; k < 4; k++) cnt[k * n_blocks + b] = c[k];
    }
}

// final layout: block b = {4 x u64 exclusive prefix counts, up to 8 words}; trailing totals
__global__ void k_interleave(const uint32_t *words, const uint64_t *excl /* 4 x n_blocks */, const uint64_t *total, uint64_t N,
                             uint64_t n_blocks, uint32_t *out)
{
    const uint64_t n_words = (N + 15) / 16;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t *o = out + b * 16;
        if (b == n_blocks) { // trailing occ block sits right after the last (possibly short) block
            o = out + n_blocks * 8 + n_words;
            for (int k = 0; k < 4; k++) { o[2 * k] = (uint32_t)total[k]; o[2 * k + 1] = (uint32_t)(total[k] >> 32); }
            continue;
        }
        for (int k = 0; k < 4; k++) { uint64_t v = excl[k * n_blocks + b]; o[2 * k] = (uint32_t)v; o[2 * k + 1] = (uint32_t)(v >> 32); }
        for (int j = 0; j < 8; j++) { uint64_t w = b * 8 + j; if (w < n_words) o[8 + j] = words[w]; }
    }
}

__global__ void k_sample_sa(const uint64_t *idx, uint64_t N, int intv, uint64_t n_sa, uint64_t *sa, uint64_t *sa_full)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= N; r += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t v = r == 0 ? N : idx[r - 1];
        if (sa_full) sa_full[r] = r == 0 ? ~0ull : v;
        if (r % intv == 0 && r / intv < n_sa) sa[r / intv] = r == 0 ? ~0ull : v;
    }
}

__global__ void k_find_primary(const uint64_t *idx, uint64_t N, unsigned long long *primary)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += (uint64_t)gridDim.x * blockDim.x)
        if (idx[j] == 0) *primary = j + 1;
}

__global__ void k_count_bases(const uint8_t *T, uint64_t N, unsigned long long *cnt)
{
    unsigned long long c[4] = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (uint64_t)gridDim.x * blockDim.x) c[T[i] & 3]++;
    for (int k = 0; k < 4; k++) {
        unsigned long long v = c[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(cnt + k, v);
    }
}

struct MaxOp { __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } };

// ---------------------------------------------------------------------------------------------
// builder: forward codes in HBM -> HostIndex-compatible device arrays
// ---------------------------------------------------------------------------------------------
int mcx_build_suffix_index(const uint8_t *d_fwd, uint64_t G, bool want_full_sa, DevIndexArrays &out, double *seconds)
{
    const uint64_t N = 2 * G;
    if (N == 0) return mcx_set_error(MCX_ERR_ARG, "empty genome");
    if (N >= (1ull << 33) - 16) return mcx_set_error(MCX_ERR_UNSUPPORTED, "GPU index construction handles texts below 2^33 positions (genome < 4.29 Gbp)");
    hipEvent_t e0, e1;
    HIP_TRYB(hipEventCreate(&e0)); HIP_TRYB(hipEventCreate(&e1));
    HIP_TRYB(hipEventRecord(e0));
    uint8_t *T = nullptr;
    uint64_t *sa = nullptr, *rank = nullptr, *key = nullptr, *key_alt = nullptr, *val_alt = nullptr;
    uint32_t *head = nullptr;
    unsigned long long *d_misc = nullptr; // [0] groups, [1] primary, [2..5] base counts, [8..263] bucket counts, [264..519] bucket cursors
    HIP_TRYB(hipMalloc(&T, N + 64));
    HIP_TRYB(hipMemset(T + N, 0, 64));
    HIP_TRYB(hipMalloc(&d_misc, 520 * sizeof(unsigned long long)));
    HIP_TRYB(hipMemset(d_misc, 0, 520 * sizeof(unsigned long long)));
    const unsigned grid = 8192, block = 256;
    k_make_text<<<grid, block>>>(d_fwd, G, T);
    k_count_bases<<<grid, block>>>(T, N, d_misc + 2);
    // buckets: one for small texts, else by the first kb bases so that every bucket stays below 2^31 suffixes
    Buckets bk;
    bk.kb = N < (1ull << 31) ? 0 : 2;
    if (const char *e = getenv("MCX_BUILD_BUCKET_BASES")) { const int k = atoi(e); if (k >= 0 && k <= 4) bk.kb = k; } // (tests exercise the bucket path on small genomes)
    unsigned long long h_count[256];
    for (;; bk.kb++) {
        if (bk.kb > 4) return mcx_set_error(MCX_ERR_UNSUPPORTED, "index construction: a 4-base bucket exceeds 2^31 suffixes");
        bk.n = 1 << (2 * bk.kb);
        HIP_TRYB(hipMemset(d_misc + 8, 0, 512 * sizeof(unsigned long long)));
        k_bucket_count<<<grid, block>>>(T, N, bk.kb, d_misc + 8);
        HIP_TRYB(hipMemcpy(h_count, d_misc + 8, sizeof h_count, hipMemcpyDeviceToHost));
        unsigned long long most = 0;
        for (int k = 0; k < bk.n; k++) most = std::max(most, h_count[k]);
        if (most < (1ull << 31)) break;
    }
    uint64_t max_nb = 0;
    bk.base[0] = 0;
    for (int k = 0; k < bk.n; k++) { bk.base[k + 1] = bk.base[k] + h_count[k]; max_nb = std::max<uint64_t>(max_nb, h_count[k]); }
    HIP_TRYB(hipMalloc(&sa, N * 8)); HIP_TRYB(hipMalloc(&rank, N * 8)); HIP_TRYB(hipMalloc(&key, N * 8));
    HIP_TRYB(hipMalloc(&key_alt, max_nb * 8)); HIP_TRYB(hipMalloc(&val_alt, max_nb * 8)); HIP_TRYB(hipMalloc(&head, max_nb * 4));
    {
        unsigned long long cur[256];
        for (int k = 0; k < 256; k++) cur[k] = k < bk.n ? bk.base[k] : 0;
        HIP_TRYB(hipMemcpy(d_misc + 264, cur, sizeof cur, hipMemcpyHostToDevice));
        k_bucket_scatter<<<grid, block>>>(T, N, bk.kb, d_misc + 264, sa);
    }
    // temp storage for sort + scan (sized for the largest bucket)
    size_t sort_bytes = 0, scan_bytes = 0;
    {
        hipcub::DoubleBuffer<uint64_t> dk(key, key_alt);
        hipcub::DoubleBuffer<uint64_t> dv(sa, val_alt);
        HIP_TRYB(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, dk, dv, (int64_t)max_nb, 0, 64));
        HIP_TRYB(hipcub::DeviceScan::InclusiveScan(nullptr, scan_bytes, head, head, MaxOp(), (int64_t)max_nb));
        size_t sum_bytes = 0;
        uint64_t *dummy = key;
        HIP_TRYB(hipcub::DeviceScan::ExclusiveSum(nullptr, sum_bytes, dummy, dummy, (int64_t)((N + 127) / 128)));
        if (sum_bytes > scan_bytes) scan_bytes = sum_bytes;
    }
    void *tmp = nullptr;
    const size_t tmp_bytes = (sort_bytes > scan_bytes ? sort_bytes : scan_bytes) + 256;
    HIP_TRYB(hipMalloc(&tmp, tmp_bytes));
    std::vector<bool> done((size_t)bk.n, false); // buckets whose suffixes are all distinct already
    std::vector<unsigned long long> groups_of((size_t)bk.n, 0);
    uint64_t h = 16;
    for (int iter = 0;; iter++) {
        // keys of every unfinished bucket first: they read the ranks of the previous round
        for (int k = 0; k < bk.n; k++) {
            const uint64_t nb = bk.base[k + 1] - bk.base[k];
            if (nb == 0 || done[k]) continue;
            if (iter == 0) k_init_keys<<<grid, block>>>(T, N, sa + bk.base[k], nb, key + bk.base[k]);
            else k_pair_keys<<<grid, block>>>(sa + bk.base[k], rank, nb, bk.base[k], N, h, key + bk.base[k]);
        }
        if (iter > 0) h <<= 1;
        unsigned long long total_groups = 0;
        for (int k = 0; k < bk.n; k++) {
            const uint64_t nb = bk.base[k + 1] - bk.base[k], b0 = bk.base[k];
            if (nb == 0) continue;
            if (done[k]) { total_groups += nb; continue; }
            int local_bits = 1;
            while ((1ull << local_bits) < nb + 2) local_bits++;
            hipcub::DoubleBuffer<uint64_t> dk(key + b0, key_alt);
            hipcub::DoubleBuffer<uint64_t> dv(sa + b0, val_alt);
            size_t sb = tmp_bytes;
            HIP_TRYB(hipcub::DeviceRadixSort::SortPairs(tmp, sb, dk, dv, (int64_t)nb, 0, iter == 0 ? 37 : 33 + local_bits));
            if (dk.Current() != key + b0) HIP_TRYB(hipMemcpyAsync(key + b0, key_alt, nb * 8, hipMemcpyDeviceToDevice));
            if (dv.Current() != sa + b0) HIP_TRYB(hipMemcpyAsync(sa + b0, val_alt, nb * 8, hipMemcpyDeviceToDevice));
            k_mark_heads<<<grid, block>>>(key + b0, nb, head);
            size_t cb = tmp_bytes;
            HIP_TRYB(hipcub::DeviceScan::InclusiveScan(tmp, cb, head, head, MaxOp(), (int64_t)nb));
            HIP_TRYB(hipMemset(d_misc, 0, sizeof(unsigned long long)));
            k_scatter_rank<<<grid, block>>>(sa + b0, head, nb, b0, rank, d_misc);
            unsigned long long groups = 0;
            HIP_TRYB(hipMemcpy(&groups, d_misc, sizeof groups, hipMemcpyDeviceToHost));
            groups_of[k] = groups;
            total_groups += groups;
        }
        // a bucket is finished once its ranks are all distinct; marked after the round so that this
        // round's keys of the other buckets saw consistent ranks
        for (int k = 0; k < bk.n; k++) if (!done[k] && groups_of[k] == bk.base[k + 1] - bk.base[k]) done[k] = true;
        if (total_groups == N) break;
        if (h >= 2 * N) return mcx_set_error(MCX_ERR_DEVICE, "suffix sorting did not converge");
    }
    // sa is the suffix array of T (without the terminator row)
    const uint64_t *SA = sa;
    k_find_primary<<<grid, block>>>(SA, N, d_misc + 1);
    unsigned long long misc[8];
    HIP_TRYB(hipMemcpy(misc, d_misc, sizeof misc, hipMemcpyDeviceToHost));
    out.primary = misc[1]; out.seq_len = N;
    out.L2[0] = 0;
    for (int k = 0; k < 4; k++) out.L2[k + 1] = out.L2[k] + misc[2 + k];
    // BWT words (the key / rank buffers are scratch now)
    const uint64_t n_words = (N + 15) / 16, n_blocks = (N + 127) / 128;
    uint32_t *words = (uint32_t *)key;
    k_pack_bwt<<<grid, block>>>(T, SA, N, out.primary, words);
    uint64_t *cnt = rank; // 4 x n_blocks counts, then exclusive sums in place
    k_block_counts<<<grid, block>>>(words, N, n_blocks, cnt);
    uint64_t *d_total = nullptr;
    HIP_TRYB(hipMalloc(&d_total, 4 * 8));
    for (int k = 0; k < 4; k++) {
        size_t cb = 0;
        HIP_TRYB(hipcub::DeviceScan::ExclusiveSum(nullptr, cb, cnt + k * n_blocks, cnt + k * n_blocks, (int64_t)n_blocks));
        if (cb > tmp_bytes) return mcx_set_error(MCX_ERR_DEVICE, "scan scratch too small");
        HIP_TRYB(hipcub::DeviceScan::ExclusiveSum(tmp, cb, cnt + k * n_blocks, cnt + k * n_blocks, (int64_t)n_blocks));
    }
    uint64_t totals[4];
    for (int k = 0; k < 4; k++) totals[k] = misc[2 + k];
    HIP_TRYB(hipMemcpy(d_total, totals, sizeof totals, hipMemcpyHostToDevice));
    out.bwt_words = n_blocks * 8 + n_words + 8;
    HIP_TRYB(hipMalloc(&out.bwt, out.bwt_words * 4 + 128));
    HIP_TRYB(hipMemset(out.bwt, 0, out.bwt_words * 4 + 128));
    k_interleave<<<grid, block>>>(words, cnt, d_total, N, n_blocks, out.bwt);
    HIP_TRYB(hipDeviceSynchronize());
    (void)hipFree(key); (void)hipFree(rank); (void)hipFree(key_alt); (void)hipFree(val_alt); (void)hipFree(head); // room for the full suffix array
    out.n_sa = (N + 32) / 32;
    HIP_TRYB(hipMalloc(&out.sa, out.n_sa * 8));
    if (want_full_sa) HIP_TRYB(hipMalloc(&out.sa_full, (N + 1) * 8 + 16)); // (+16: rows are fetched in pairs, seed_take)
    k_sample_sa<<<grid, block>>>(SA, N, 32, out.n_sa, out.sa, out.sa_full);
    HIP_TRYB(hipGetLastError());
    HIP_TRYB(hipEventRecord(e1));
    HIP_TRYB(hipDeviceSynchronize());
    float ms = 0;
    HIP_TRYB(hipEventElapsedTime(&ms, e0, e1));
    if (seconds) *seconds = ms / 1000.0;
    (void)hipFree(T); (void)hipFree(sa); (void)hipFree(d_misc); (void)hipFree(tmp); (void)hipFree(d_total);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// pair records (mcx_fm.h PairSlot): the two bases before every suffix, counted per 32 BWT symbols
// ---------------------------------------------------------------------------------------------
// one thread per stored symbol: its code (a byte: the records are made from these in a second pass, once the counts before
// every workgroup's 256 symbols are known) and the workgroup's count of every code
__global__ void __launch_bounds__(256) k_pair_codes(IndexView ix, uint8_t *codes, uint64_t *cnt, uint64_t n_groups, unsigned long long *lone)
{
    __shared__ uint32_t h[16];
    for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) { // (a launch holds fewer than 2^32 threads: the groups in a loop)
        if (threadIdx.x < 16) h[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t i = g * 256 + threadIdx.x;
        bool ln;
        const int code = fm_pair_code(ix, i, ln);
        if (ln) *lone = i;
        codes[i] = (uint8_t)code;
        for (int j = 0; j < 16; j++) {
            const uint64_t m = __ballot(code == j);
            if ((threadIdx.x & 63) == 0 && m) atomicAdd(&h[j], (uint32_t)__popcll(m));
        }
        __syncthreads();
        if (threadIdx.x < 16) cnt[(uint64_t)threadIdx.x * n_groups + g] = h[threadIdx.x];
        __syncthreads();
    }
}

// cnt: per code the exclusive sums over the workgroups.  A wavefront makes the records of two runs of 32 symbols: lane j (1..15)
// holds "code below j" of both as ballots, and writes slot j - 1 (j <= 8) and slot j (j >= 8) of either record
struct PairStarts { uint64_t s[16]; };
__global__ void __launch_bounds__(256) k_pair_records(const uint8_t *codes, const uint64_t *cnt, PairStarts st, uint64_t n_groups, PairSlot *rec)
{
    __shared__ uint32_t part[8][16]; // symbols with a code below j in each of the group's eight runs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint64_t i = g * 256 + threadIdx.x;
        const int code = codes[i];
        uint32_t lt0 = 0, lt1 = 0;
        for (int j = 1; j < 16; j++) {
            const uint64_t m = __ballot(code < j); // (kPairNone is below nothing)
            if (lane == j) { lt0 = __brev((uint32_t)m); lt1 = __brev((uint32_t)(m >> 32)); } // symbol t of a run at bit 31 - t
        }
        if (lane >= 1 && lane < 16) { part[2 * wave][lane] = (uint32_t)__popc(lt0); part[2 * wave + 1][lane] = (uint32_t)__popc(lt1); }
        __syncthreads();
        if (lane >= 1 && lane < 16) {
            uint64_t before = 0;
            for (int j = 0; j < lane; j++) before += cnt[(uint64_t)j * n_groups + g] - st.s[j]; // (one scan over the arrays laid end to end: a code's sums start at the total of the codes below it)
            for (int r = 0; r < 2 * wave; r++) before += part[r][lane];
            PairSlot *out = rec + (g * 8 + 2 * wave) * 16;
            PairSlot e0, e1;
            e0.lt = lt0; e0.n_lt = (uint32_t)before;
            e1.lt = lt1; e1.n_lt = (uint32_t)(before + part[2 * wave][lane]);
            if (lane <= 8) { out[lane - 1] = e0; out[16 + lane - 1] = e1; }
            if (lane >= 8) { out[lane] = e0; out[16 + lane] = e1; }
        }
        __syncthreads();
    }
}

__global__ void k_pair_first(IndexView ix, uint64_t *c2)
{
    if (threadIdx.x < 16) c2[threadIdx.x] = fm_pair_first(ix, (int)threadIdx.x);
    if (threadIdx.x == 16) c2[16] = (uint64_t)ref_code(ix, 0);
}

__global__ void k_pair_check(IndexView ix, uint64_t trials, unsigned long long *bad)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < trials && !fm_pair_step_agrees(ix, t)) atomicAdd(bad, 1ull);
}

int mcx_build_pair_records(IndexView &v, void **d_rec, void **d_c2, int64_t *bytes, int check_trials)
{
    *d_rec = nullptr; *d_c2 = nullptr; *bytes = 0;
    if (!v.sa_full || v.seq_len < 2) return 0;
    const uint64_t n_groups = (v.seq_len + 255) / 256, n_chunks = n_groups * 8;
    uint8_t *codes = nullptr;
    uint64_t *cnt = nullptr, *c2 = nullptr;
    unsigned long long *d_misc = nullptr;
    void *tmp = nullptr;
    PairSlot *rec = nullptr;
    auto drop = [&]() { (void)hipFree(codes); (void)hipFree(cnt); (void)hipFree(d_misc); (void)hipFree(tmp); };
    auto fail = [&](const std::string &what, hipError_t e) { drop(); (void)hipFree(rec); (void)hipFree(c2); return mcx_set_error(MCX_ERR_DEVICE, "pair records: " + what + ": " + hipGetErrorString(e)); };
    hipError_t e;
    if ((e = hipMalloc(&codes, n_groups * 256)) != hipSuccess) return fail("hipMalloc(codes)", e);
    if ((e = hipMalloc(&cnt, (16 * n_groups + 1) * 8)) != hipSuccess) return fail("hipMalloc(counts)", e);
    if ((e = hipMalloc(&d_misc, 16)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMemset(d_misc, 0xFF, 8)) != hipSuccess || (e = hipMemset(d_misc + 1, 0, 8)) != hipSuccess) return fail("hipMemset", e);
    k_pair_codes<<<(unsigned)std::min<uint64_t>(n_groups, 1u << 20), 256>>>(v, codes, cnt, n_groups, d_misc);
    if ((e = hipGetLastError()) != hipSuccess) return fail("k_pair_codes", e);
    size_t cb = 0;
    if ((e = hipcub::DeviceScan::ExclusiveSum(nullptr, cb, cnt, cnt, (int64_t)(16 * n_groups + 1))) != hipSuccess) return fail("scan size", e);
    if ((e = hipMalloc(&tmp, cb)) != hipSuccess) return fail("hipMalloc(scan)", e);
    if ((e = hipMemset(cnt + 16 * n_groups, 0, 8)) != hipSuccess) return fail("hipMemset", e);
    if ((e = hipcub::DeviceScan::ExclusiveSum(tmp, cb, cnt, cnt, (int64_t)(16 * n_groups + 1))) != hipSuccess) return fail("scan", e);
    uint64_t starts[17];
    for (int j = 0; j <= 16; j++)
        if ((e = hipMemcpy(&starts[j], cnt + (uint64_t)j * n_groups, 8, hipMemcpyDeviceToHost)) != hipSuccess) return fail("hipMemcpy", e);
    unsigned long long lone = 0;
    if ((e = hipMemcpy(&lone, d_misc, 8, hipMemcpyDeviceToHost)) != hipSuccess) return fail("hipMemcpy", e);
    for (int j = 0; j < 16; j++)
        if ((starts[j + 1] - starts[j]) >> 32) { drop(); return 0; } // a pair of bases 2^32 times or more: the walk keeps its single steps
    const size_t rec_bytes = (size_t)n_chunks * 16 * sizeof(PairSlot) + 64;
    if ((e = hipMalloc(&rec, rec_bytes)) != hipSuccess) return fail("hipMalloc(records)", e);
    if ((e = hipMalloc(&c2, 17 * 8)) != hipSuccess) return fail("hipMalloc", e);
    PairStarts st;
    for (int j = 0; j < 16; j++) st.s[j] = starts[j];
    k_pair_records<<<(unsigned)std::min<uint64_t>(n_groups, 1u << 20), 256>>>(codes, cnt, st, n_groups, rec);
    if ((e = hipGetLastError()) != hipSuccess) return fail("k_pair_records", e);
    k_pair_first<<<1, 64>>>(v, c2);
    uint64_t t0 = 0;
    if ((e = hipMemcpy(&t0, c2 + 16, 8, hipMemcpyDeviceToHost)) != hipSuccess) return fail("k_pair_first", e);
    drop(); codes = nullptr; cnt = nullptr; d_misc = nullptr; tmp = nullptr;
    v.rank2 = rec; v.rank2_c2 = c2; v.rank2_lone = lone; v.rank2_t0 = (int32_t)t0;
    if (check_trials > 0) {
        unsigned long long bad = 0, *d_bad = nullptr;
        if ((e = hipMalloc(&d_bad, 8)) == hipSuccess) e = hipMemset(d_bad, 0, 8);
        if (e == hipSuccess) {
            k_pair_check<<<(unsigned)(((uint64_t)check_trials + 255) / 256), 256>>>(v, (uint64_t)check_trials, d_bad);
            e = hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
        }
        (void)hipFree(d_bad);
        if (e != hipSuccess || bad) {
            v.rank2 = nullptr; v.rank2_c2 = nullptr;
            (void)hipFree(rec); (void)hipFree(c2);
            if (e != hipSuccess) return mcx_set_error(MCX_ERR_DEVICE, std::string("pair records: check: ") + hipGetErrorString(e));
            return mcx_set_error(MCX_ERR_DEVICE, "pair records: " + std::to_string(bad) + " of " + std::to_string(check_trials) + " two-base steps differ from two single steps");
        }
    }
    *d_rec = rec; *d_c2 = c2; *bytes = (int64_t)rec_bytes + 17 * 8;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// FASTA -> packed genome + .ann/.amb/.pac exactly as bns_fasta2bntseq(fp, prefix, for_only=1)
// ---------------------------------------------------------------------------------------------
struct FastaSeq { std::string name, comment, seq; };

static bool read_fasta(const char *path, std::vector<FastaSeq> &out, std::string &err)
{
    gzFile g = gzopen(path, "rb");
    if (!g) { err = std::string("cannot open ") + path; return false; }
    gzbuffer(g, 1 << 20);
    std::vector<char> buf(1 << 16);
    std::string line;
    FastaSeq *cur = nullptr;
    for (;;) {
        line.clear();
        bool got = false;
        while (gzgets(g, buf.data(), (int)buf.size())) { got = true; line += buf.data(); if (!line.empty() && line.back() == '\n') break; }
        if (!got) break;
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (!line.empty() && line[0] == '>') {
            out.emplace_back();
            cur = &out.back();
            size_t i = 1;
            while (i < line.size() && !isspace((unsigned char)line[i])) i++;
            cur->name = line.substr(1, i - 1);
            cur->comment = i < line.size() ? line.substr(i + 1) : std::string(); // kseq: the rest after one delimiter
        } else if (cur) {
            for (char c : line) if (!isspace((unsigned char)c)) cur->seq.push_back(c); // kseq keeps graphic characters only
        }
    }
    gzclose(g);
    if (out.empty()) { err = std::string(path) + " holds no FASTA record"; return false; }
    return true;
}

extern "C" int mcx_index_build(const char *fasta_path, const char *prefix, int device)
{
    if (!fasta_path || !prefix) return mcx_set_error(MCX_ERR_ARG, "mcx_index_build: null argument");
    std::vector<FastaSeq> seqs;
    std::string err;
    if (!read_fasta(fasta_path, seqs, err)) return mcx_set_error(MCX_ERR_IO, err);
    HIP_TRYB(hipSetDevice(device));
    // pack: N (anything that is not ACGT) -> lrand48() & 3 with the generator seeded by 11 (bntseq.c:171-172, :130)
    srand48(11);
    struct Hole { long long off; int len; char amb; };
    std::vector<Hole> holes;
    std::vector<uint8_t> codes;
    struct Ann { long long off; int len, n_ambs; };
    std::vector<Ann> anns;
    long long l_pac = 0;
    for (auto &s : seqs) {
        Ann a; a.off = l_pac; a.len = (int)s.seq.size(); a.n_ambs = 0;
        int lasts = 0;
        for (size_t i = 0; i < s.seq.size(); i++) {
            int ch = (unsigned char)s.seq[i];
            int c;
            switch (ch) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break; default: c = 4; }
            if (c >= 4) {
                if (lasts == ch) holes.back().len++;
                else { Hole h; h.off = a.off + (long long)i; h.len = 1; h.amb = (char)ch; holes.push_back(h); a.n_ambs++; }
                c = (int)(lrand48() & 3);
            }
            lasts = ch;
            codes.push_back((uint8_t)c);
            l_pac++;
        }
        anns.push_back(a);
    }
    const uint64_t G = (uint64_t)l_pac;
    std::string p(prefix);
    { // .pac (bntseq.c:204-218)
        std::vector<uint8_t> pac(G / 4 + 2, 0);
        for (uint64_t i = 0; i < G; i++) pac[i >> 2] |= codes[i] << ((~i & 3) << 1);
        FILE *f = fopen((p + ".pac").c_str(), "wb");
        if (!f) return mcx_set_error(MCX_ERR_IO, "cannot write " + p + ".pac");
        size_t nb = (G >> 2) + ((G & 3) == 0 ? 0 : 1);
        fwrite(pac.data(), 1, nb, f);
        uint8_t ct = 0;
        if (G % 4 == 0) fwrite(&ct, 1, 1, f);
        ct = (uint8_t)(G % 4);
        fwrite(&ct, 1, 1, f);
        fclose(f);
    }
    { // .ann / .amb (bns_dump, bntseq.c:60-91)
        FILE *f = fopen((p + ".ann").c_str(), "w");
        if (!f) return mcx_set_error(MCX_ERR_IO, "cannot write " + p + ".ann");
        fprintf(f, "%lld %d %u\n", l_pac, (int)seqs.size(), 11u);
        for (size_t i = 0; i < seqs.size(); i++) {
            fprintf(f, "%d %s", 0, seqs[i].name.c_str());
            const std::string anno = seqs[i].comment.empty() ? std::string("(null)") : seqs[i].comment;
            fprintf(f, " %s\n", anno.c_str());
            fprintf(f, "%lld %d %d\n", anns[i].off, anns[i].len, anns[i].n_ambs);
        }
        fclose(f);
        f = fopen((p + ".amb").c_str(), "w");
        if (!f) return mcx_set_error(MCX_ERR_IO, "cannot write " + p + ".amb");
        fprintf(f, "%lld %d %u\n", l_pac, (int)seqs.size(), (unsigned)holes.size());
        for (auto &h : holes) fprintf(f, "%lld %d %c\n", h.off, h.len, h.amb);
        fclose(f);
    }
    uint8_t *d_fwd = nullptr;
    HIP_TRYB(hipMalloc(&d_fwd, G + 64));
    HIP_TRYB(hipMemcpy(d_fwd, codes.data(), G, hipMemcpyHostToDevice));
    DevIndexArrays arr;
    int rc = mcx_build_suffix_index(d_fwd, G, false, arr, nullptr);
    (void)hipFree(d_fwd);
    if (rc) return rc;
    // .bwt: primary, L2[1..4], words (bwt_dump_bwt, bwt.c:174-184); .sa (bwt_dump_sa :186-198)
    const uint64_t file_words = arr.bwt_words; // n_blocks*8 + n_words + 8
    std::vector<uint32_t> words(file_words);
    HIP_TRYB(hipMemcpy(words.data(), arr.bwt, file_words * 4, hipMemcpyDeviceToHost));
    std::vector<uint64_t> sa(arr.n_sa);
    HIP_TRYB(hipMemcpy(sa.data(), arr.sa, arr.n_sa * 8, hipMemcpyDeviceToHost));
    (void)hipFree(arr.bwt); (void)hipFree(arr.sa);
    FILE *f = fopen((p + ".bwt").c_str(), "wb");
    if (!f) return mcx_set_error(MCX_ERR_IO, "cannot write " + p + ".bwt");
    fwrite(&arr.primary, 8, 1, f); fwrite(arr.L2 + 1, 8, 4, f); fwrite(words.data(), 4, file_words, f);
    fclose(f);
    f = fopen((p + ".sa").c_str(), "wb");
    if (!f) return mcx_set_error(MCX_ERR_IO, "cannot write " + p + ".sa");
    const uint64_t intv = 32;
    fwrite(&arr.primary, 8, 1, f); fwrite(arr.L2 + 1, 8, 4, f); fwrite(&intv, 8, 1, f); fwrite(&arr.seq_len, 8, 1, f);
    fwrite(sa.data() + 1, 8, arr.n_sa - 1, f);
    fclose(f);
    return 0;
}
