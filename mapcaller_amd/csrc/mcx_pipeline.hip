// mapcaller_amd/csrc/mcx_pipeline.hip — HIP kernels, the batch pipeline and the C ABI.
//
// One batch of reads goes through (all on one stream, no host round trip inside a tier):
//   k_pack_reads  ASCII -> 2-bit words + N masks, mate 2 reverse-complemented (ReadMapping.cpp:447,451)
//   k_seed        jump table + FM steps + direct comparison, one read per lane (IdentifySimplePairs/BWT_Search)
//   k_sa          hits that are still BWT rows -> text positions            (bwt_sa)
//   k_cluster     one pair per lane: sort, cluster, pair by distance        (SimplePairClustering, CheckPairedAlignmentDistance)
//   k_rescue      unpaired pairs only, one workgroup each: 8-mer mate rescue (AlignmentRescue)
//   k_build       mask, fragment lists, DP job emission by size class       (ProduceReadAlignment up to ProcessNormalPair)
//   k_dp_small / k_dp_sel<K>  16-lane groups / one wavefront per DP job     (ksw2_alignment / nw_alignment)
//   k_finish      gates, scores, pair stats, flags, MAPQ, CIGAR, records    (ProduceReadAlignment tail, SamReport.cpp)
// Pairs that overflow the tier-0 pair-state capacities are re-run in tier 1 (hard bounds).
// The reference's per-chunk avgDist feedback is then replayed: per-chunk sums on the device, the
// trajectory on the host, and only the pairs whose decision depends on the exact estimate re-run.
// The -vcf bookkeeping of a batch (mcx_profile.h) follows when a profile is attached.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/mcx.h"
#include "mcx_dp.h"
#include "mcx_dp_lane2.h"
#include "mcx_simple.h"
#include "mcx_profile.h"
#include <hipcub/hipcub.hpp>
#include "mcx_internal.h"

using namespace mcx;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(MCX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

int mcx_set_error(int code, const std::string &msg) { return fail(code, msg); }
extern "C" const char *mcx_last_error(void) { return g_err.c_str(); }
extern "C" int mcx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------------------------
// index
// ---------------------------------------------------------------------------------------------

// Expands the sampled suffix array: the chain of LF steps that starts at a sampled row visits
// exactly the rows whose bwt_sa() walk ends at the next sampled row, with values one lower per
// step (SA[LF(k)] = SA[k] - 1).  One chain per lane, ~32 dependent block fetches each.
__global__ void k_expand_sa(IndexView ix, uint64_t n_sa, uint64_t *full)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_sa) return;
    uint64_t k = j * (uint64_t)ix.sa_intv;
    uint64_t val = j == 0 ? ix.seq_len : ix.sa[j];
    full[k] = j == 0 ? ~0ull : val;
    const uint64_t mask = (uint64_t)ix.sa_intv - 1;
    for (;;) {
        k = fm_lf(ix, k);
        val -= 1;
        if ((k & mask) == 0) break;
        full[k] = val;
    }
}

// the derived form of the index blocks (mcx_fm.h fm_derive_block): one thread per block that holds symbols
__global__ void k_derive_bwt(uint32_t *bwt, uint64_t n_blocks)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_blocks) fm_derive_block(bwt + (i << 4));
}

static int derive_bwt(mcx_index *ix)
{
    const uint64_t n_blocks = (ix->host.seq_len + 127) / 128;
    k_derive_bwt<<<(unsigned)((n_blocks + 255) / 256), 256>>>((uint32_t *)ix->d_bwt, n_blocks);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

__global__ void k_build_ktab(IndexView ix, int K, U4 *tab)
{
    const uint64_t n = 1ull << (2 * K);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x0, x1, x2;
        ktab_entry(ix, (uint32_t)i, K, x0, x1, x2);
        tab[i] = ktab_pack(x0, x1, x2);
    }
}

// the K-mer jump table of the seeding walk (mcx_fm.h): 16 bytes per K-mer, K from the text length
// (MCX_KTAB_K overrides it for experiments)
static int build_rank(mcx_index *ix);
static int build_ktab(mcx_index *ix)
{
    int K = ktab_k_for(ix->view.seq_len);
    if (const char *e = getenv("MCX_KTAB_K")) { const int k = atoi(e); if (k >= 4 && k <= 16) K = k; } // (16: 69 GB — one pair step fewer per search; no room for it beside the -vcf planes)
    const size_t bytes = (size_t)16 << (2 * K);
    hipError_t e = hipMalloc(&ix->d_ktab, bytes);
    if (e != hipSuccess) { g_err = std::string("hipMalloc(ktab): ") + hipGetErrorString(e); return MCX_ERR_DEVICE; }
    ix->view.ktab = nullptr; ix->view.ktab_k = K;
    k_build_ktab<<<8192, 256>>>(ix->view, K, (U4 *)ix->d_ktab);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { g_err = std::string("k_build_ktab: ") + hipGetErrorString(e); return MCX_ERR_DEVICE; }
    ix->view.ktab = (const uint32_t *)ix->d_ktab;
    ix->hbm_bytes += (int64_t)bytes;
    return build_rank(ix);
}

// the rank records of the seeding walk (mcx_fm.h RankChunk): one thread per .bwt block, four records per base
__global__ void k_build_rank(const uint32_t *bwt, uint64_t n_blocks, uint64_t n_chunks, RankChunk *rank, unsigned long long *cross)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_blocks) return;
    const uint32_t *blk = bwt + (i << 4);
    uint64_t before[4];
    for (int c = 0; c < 4; c++) before[c] = fm_plain_count((uint64_t)blk[2 * c] | ((uint64_t)blk[2 * c + 1] << 32));
    for (int q = 0; q < 4; q++) {
        const uint64_t chunk = 4 * i + q;
        if (chunk >= n_chunks) break;
        RankChunk out[4];
        uint64_t ne[4], ng[4];
        fm_rank_records(blk[8 + 2 * q], blk[9 + 2 * q], before, out, ne, ng);
        for (int b = 0; b < 4; b++) {
            rank[(uint64_t)b * n_chunks + chunk] = out[b];
            if (ne[b] >> 32) atomicMin(&cross[b], (unsigned long long)chunk);
            if (ng[b] >> 32) atomicMin(&cross[4 + b], (unsigned long long)chunk);
            before[b] += (uint64_t)__popc(out[b].eq);
        }
    }
}

static int build_rank(mcx_index *ix)
{
    if (!ix->view.sa_full || getenv("MCX_NO_RANK")) return 0; // (the walk then counts in the .bwt blocks: MCX_NO_RANK for experiments)
    // a record keeps its two running counts in 32 bits plus ONE crossing chunk per base and count (rank_cross): exact while no count
    // passes 2^32 twice, i.e. below 2^33 symbols.  Longer texts (genomes above ~4.29 Gbp) walk the .bwt blocks, which have no such limit.
    if (ix->host.seq_len >= ((uint64_t)1 << 33)) return 0;
    const uint64_t n_blocks = (ix->host.seq_len + 127) / 128, n_chunks = (ix->host.seq_len + 31) / 32;
    const size_t bytes = (size_t)4 * n_chunks * sizeof(RankChunk) + 64;
    unsigned long long *d_cross = nullptr, h_cross[8];
    for (auto &x : h_cross) x = ~0ull;
    hipError_t e = hipMalloc(&ix->d_rank, bytes);
    if (e != hipSuccess) { g_err = std::string("hipMalloc(rank records): ") + hipGetErrorString(e); return MCX_ERR_DEVICE; }
    HIP_TRY(hipMalloc((void **)&d_cross, sizeof h_cross));
    e = hipMemcpy(d_cross, h_cross, sizeof h_cross, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_build_rank<<<(unsigned)((n_blocks + 255) / 256), 256>>>((const uint32_t *)ix->d_bwt, n_blocks, n_chunks, (RankChunk *)ix->d_rank, d_cross);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(h_cross, d_cross, sizeof h_cross, hipMemcpyDeviceToHost);
    (void)hipFree(d_cross);
    if (e != hipSuccess) return fail(MCX_ERR_DEVICE, std::string("rank records: ") + hipGetErrorString(e)); // (d_rank is the index's: mcx_index_free releases it)
    ix->view.rank = ix->d_rank; ix->view.rank_chunks = n_chunks;
    for (int k = 0; k < 8; k++) ix->view.rank_cross[k] = h_cross[k];
    ix->hbm_bytes += (int64_t)bytes;
    // the pair records on top (two bases per step): MCX_NO_RANK2 for experiments, MCX_RANK2_CHECK=n extends n random intervals both ways
    if (!ix->pair_records || getenv("MCX_NO_RANK2")) return 0;
    const char *chk = getenv("MCX_RANK2_CHECK");
    const int rc = mcx_build_pair_records(ix->view, &ix->d_rank2, &ix->d_rank2_c2, &ix->rank2_bytes, chk ? atoi(chk) : 0);
    if (rc && ix->pair_records == MCX_INDEX_PAIRS_IF_ROOM) {
        // nobody asked for the records by name (the CLI without -vcf takes them when there is room): a device that is too full for them
        // — several shards on it, a smaller part — keeps the one-base walk, which needs nothing more
        fprintf(stderr, "[mcx] the pair records do not fit this device (%s): the seeding walk takes one base per step\n", g_err.c_str());
        (void)hipGetLastError();
        g_err.clear();
        ix->d_rank2 = ix->d_rank2_c2 = nullptr; ix->rank2_bytes = 0;
        ix->view.rank2 = nullptr; ix->view.rank2_c2 = nullptr;
        return 0;
    }
    if (rc) return rc;
    ix->hbm_bytes += ix->rank2_bytes;
    return 0;
}

static int upload(void **dst, const void *src, size_t bytes, size_t pad, int64_t &acc)
{
    HIP_TRY(hipMalloc(dst, bytes + pad));
    if (pad) HIP_TRY(hipMemset((uint8_t *)*dst + bytes, 0, pad));
    HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    acc += (int64_t)(bytes + pad);
    return 0;
}

static int index_to_device(mcx_index *ix, int full_sa)
{
    HostIndex &h = ix->host;
    int rc;
    if ((rc = upload(&ix->d_bwt, h.bwt.data(), h.bwt.size() * 4, 128, ix->hbm_bytes))) return rc;
    if ((rc = upload(&ix->d_sa, h.sa.data(), h.sa.size() * 8, 0, ix->hbm_bytes))) return rc;
    if ((rc = upload(&ix->d_pac, h.pac.data(), h.pac.size(), 16, ix->hbm_bytes))) return rc;
    if ((rc = upload(&ix->d_end_pos, h.end_pos.data(), h.end_pos.size() * 8, 0, ix->hbm_bytes))) return rc;
    if ((rc = upload(&ix->d_end_chr, h.end_chr.data(), h.end_chr.size() * 4, 0, ix->hbm_bytes))) return rc;
    if ((rc = upload(&ix->d_chr_fwd, h.chr_fwd.data(), h.chr_fwd.size() * 8, 0, ix->hbm_bytes))) return rc;
    IndexView &v = ix->view;
    v.bwt = (const uint32_t *)ix->d_bwt; v.sa = (const uint64_t *)ix->d_sa; v.sa_full = nullptr; v.ktab = nullptr; v.ktab_k = 0; v.rank = nullptr; v.rank_chunks = 0; for (auto &x : v.rank_cross) x = ~0ull; v.rank2 = nullptr; v.rank2_c2 = nullptr; v.rank2_lone = ~0ull; v.rank2_t0 = 0;
    v.pac = (const uint8_t *)ix->d_pac;
    v.end_pos = (const int64_t *)ix->d_end_pos; v.end_chr = (const int32_t *)ix->d_end_chr;
    v.chr_fwd = (const int64_t *)ix->d_chr_fwd;
    v.primary = h.primary; for (int i = 0; i < 5; i++) v.L2[i] = h.L2[i];
    v.seq_len = h.seq_len; v.G = h.G; v.G2 = 2 * h.G;
    v.n_ends = (int32_t)h.end_pos.size(); v.n_chr = (int32_t)h.chr_len.size(); v.sa_intv = h.sa_intv;
    if ((rc = derive_bwt(ix))) return rc;
    if (full_sa) {
        size_t bytes = (size_t)(h.seq_len + 1) * 8;
        HIP_TRY(hipMalloc(&ix->d_sa_full, bytes + 16)); // (+16: rows are fetched in pairs, seed_take)
        ix->hbm_bytes += (int64_t)bytes;
        uint64_t n_sa = h.sa.size();
        k_expand_sa<<<(unsigned)((n_sa + 255) / 256), 256>>>(v, n_sa, (uint64_t *)ix->d_sa_full);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        v.sa_full = (const uint64_t *)ix->d_sa_full;
    }
    return build_ktab(ix);
}

extern "C" int mcx_index_load(const char *prefix, int device, int full_sa, mcx_index **out)
{
    if (!prefix || !out) return fail(MCX_ERR_ARG, "mcx_index_load: null argument");
    mcx_index *ix = new mcx_index();
    std::string err;
    if (!host_index_load(prefix, ix->host, err)) { delete ix; return fail(MCX_ERR_IO, err); }
    ix->device = device;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { delete ix; return fail(MCX_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e)); }
    ix->pair_records = full_sa >= 2 ? full_sa : 0;
    int rc = index_to_device(ix, full_sa);
    if (rc) { mcx_index_free(ix); return rc; }
    *out = ix;
    return 0;
}

__global__ void k_pack_pac(const uint8_t *codes, uint64_t G, uint8_t *pac)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < G / 4 + 1; b += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) { uint64_t i = b * 4 + k; v = (v << 2) | (i < G ? (codes[i] & 3u) : 0u); }
        pac[b] = (uint8_t)v;
    }
}

static int index_from_arrays(mcx_index *ix, const DevIndexArrays &arr, const uint8_t *d_codes, int64_t G);

extern "C" int mcx_index_from_codes(const uint8_t *d_codes, int32_t n_chr, const int32_t *chr_len, const char *const *chr_name,
                                    int device, int full_sa, mcx_index **out, double *build_seconds)
{
    if (!d_codes || !chr_len || n_chr <= 0 || !out) return fail(MCX_ERR_ARG, "mcx_index_from_codes: bad argument");
    HIP_TRY(hipSetDevice(device));
    mcx_index *ix = new mcx_index();
    ix->device = device;
    HostIndex &h = ix->host;
    int64_t G = 0;
    for (int i = 0; i < n_chr; i++) {
        h.chr_len.push_back(chr_len[i]);
        h.chr_name.push_back(chr_name && chr_name[i] ? chr_name[i] : ("chr" + std::to_string(i + 1)));
        G += chr_len[i];
    }
    h.G = G;
    host_index_finish(h);
    DevIndexArrays arr;
    int rc = mcx_build_suffix_index(d_codes, (uint64_t)G, full_sa != 0, arr, build_seconds);
    if (rc) { delete ix; return rc; }
    ix->pair_records = full_sa >= 2 ? full_sa : 0;
    rc = index_from_arrays(ix, arr, d_codes, G);
    if (rc) { mcx_index_free(ix); return rc; }
    *out = ix;
    return 0;
}

static int index_from_arrays(mcx_index *ix, const DevIndexArrays &arr, const uint8_t *d_codes, int64_t G)
{
    HostIndex &h = ix->host;
    int rc;
    h.primary = arr.primary; for (int i = 0; i < 5; i++) h.L2[i] = arr.L2[i];
    h.seq_len = arr.seq_len; h.sa_intv = 32;
    ix->d_bwt = arr.bwt; ix->d_sa = arr.sa; ix->d_sa_full = arr.sa_full;
    ix->hbm_bytes = (int64_t)(arr.bwt_words * 4 + arr.n_sa * 8 + (arr.sa_full ? (arr.seq_len + 1) * 8 : 0));
    ix->n_bwt_words = arr.bwt_words; ix->n_sa = arr.n_sa;
    HIP_TRY(hipMalloc(&ix->d_pac, (size_t)G / 4 + 32));
    k_pack_pac<<<1024, 256>>>(d_codes, (uint64_t)G, (uint8_t *)ix->d_pac);
    HIP_TRY(hipGetLastError());
    int64_t acc = 0;
    if ((rc = upload(&ix->d_end_pos, h.end_pos.data(), h.end_pos.size() * 8, 0, acc))) return rc;
    if ((rc = upload(&ix->d_end_chr, h.end_chr.data(), h.end_chr.size() * 4, 0, acc))) return rc;
    if ((rc = upload(&ix->d_chr_fwd, h.chr_fwd.data(), h.chr_fwd.size() * 8, 0, acc))) return rc;
    ix->hbm_bytes += acc + G / 4 + 32;
    IndexView &v = ix->view;
    v.bwt = (const uint32_t *)ix->d_bwt; v.sa = (const uint64_t *)ix->d_sa; v.sa_full = (const uint64_t *)ix->d_sa_full; v.ktab = nullptr; v.ktab_k = 0; v.rank = nullptr; v.rank_chunks = 0; for (auto &x : v.rank_cross) x = ~0ull; v.rank2 = nullptr; v.rank2_c2 = nullptr; v.rank2_lone = ~0ull; v.rank2_t0 = 0;
    v.pac = (const uint8_t *)ix->d_pac;
    v.end_pos = (const int64_t *)ix->d_end_pos; v.end_chr = (const int32_t *)ix->d_end_chr; v.chr_fwd = (const int64_t *)ix->d_chr_fwd;
    v.primary = h.primary; for (int i = 0; i < 5; i++) v.L2[i] = h.L2[i];
    v.seq_len = h.seq_len; v.G = h.G; v.G2 = 2 * h.G;
    v.n_ends = (int32_t)h.end_pos.size(); v.n_chr = (int32_t)h.chr_len.size(); v.sa_intv = 32;
    HIP_TRY(hipDeviceSynchronize());
    if ((rc = derive_bwt(ix))) return rc;
    return build_ktab(ix);
}

// writes <prefix>.bwt/.sa/.pac/.ann/.amb from an index built in HBM (no ambiguity holes: the
// codes it was built from had none)
extern "C" int mcx_index_save(const mcx_index *ix, const char *prefix)
{
    if (!ix || !prefix) return fail(MCX_ERR_ARG, "mcx_index_save: null argument");
    if (!ix->n_bwt_words) return fail(MCX_ERR_ARG, "mcx_index_save: only indexes built with mcx_index_from_codes can be saved");
    HIP_TRY(hipSetDevice(ix->device));
    const HostIndex &h = ix->host;
    std::string p(prefix);
    std::vector<uint32_t> words(ix->n_bwt_words);
    HIP_TRY(hipMemcpy(words.data(), ix->d_bwt, words.size() * 4, hipMemcpyDeviceToHost));
    { // the file holds the plain counts (the blocks in HBM carry sub-block counts in their top bits: fm_derive_block)
        const uint64_t n_blocks = (h.seq_len + 127) / 128;
        for (uint64_t i = 0; i < n_blocks && i * 16 + 8 <= words.size(); i++) for (int x = 0; x < 4; x++) words[i * 16 + 2 * x + 1] &= 0xFFu;
    }
    FILE *f = fopen((p + ".bwt").c_str(), "wb");
    if (!f) return fail(MCX_ERR_IO, "cannot write " + p + ".bwt");
    fwrite(&h.primary, 8, 1, f); fwrite(h.L2 + 1, 8, 4, f); fwrite(words.data(), 4, words.size(), f); fclose(f);
    std::vector<uint64_t> sa(ix->n_sa);
    HIP_TRY(hipMemcpy(sa.data(), ix->d_sa, sa.size() * 8, hipMemcpyDeviceToHost));
    f = fopen((p + ".sa").c_str(), "wb");
    if (!f) return fail(MCX_ERR_IO, "cannot write " + p + ".sa");
    const uint64_t intv = 32;
    fwrite(&h.primary, 8, 1, f); fwrite(h.L2 + 1, 8, 4, f); fwrite(&intv, 8, 1, f); fwrite(&h.seq_len, 8, 1, f);
    fwrite(sa.data() + 1, 8, sa.size() - 1, f); fclose(f);
    const uint64_t G = (uint64_t)h.G;
    std::vector<uint8_t> pac(G / 4 + 1);
    HIP_TRY(hipMemcpy(pac.data(), ix->d_pac, pac.size(), hipMemcpyDeviceToHost));
    f = fopen((p + ".pac").c_str(), "wb");
    if (!f) return fail(MCX_ERR_IO, "cannot write " + p + ".pac");
    fwrite(pac.data(), 1, (G >> 2) + ((G & 3) == 0 ? 0 : 1), f);
    uint8_t ct = 0;
    if (G % 4 == 0) fwrite(&ct, 1, 1, f);
    ct = (uint8_t)(G % 4); fwrite(&ct, 1, 1, f); fclose(f);
    f = fopen((p + ".ann").c_str(), "w");
    if (!f) return fail(MCX_ERR_IO, "cannot write " + p + ".ann");
    fprintf(f, "%lld %d %u\n", (long long)h.G, (int)h.chr_len.size(), 11u);
    for (size_t i = 0; i < h.chr_len.size(); i++)
        fprintf(f, "%d %s (null)\n%lld %d %d\n", 0, h.chr_name[i].c_str(), (long long)h.chr_fwd[i], h.chr_len[i], 0);
    fclose(f);
    f = fopen((p + ".amb").c_str(), "w");
    if (!f) return fail(MCX_ERR_IO, "cannot write " + p + ".amb");
    fprintf(f, "%lld %d %u\n", (long long)h.G, (int)h.chr_len.size(), 0u);
    fclose(f);
    return 0;
}

extern "C" void mcx_index_free(mcx_index *ix)
{
    if (!ix) return;
    void *p[] = {ix->d_bwt, ix->d_sa, ix->d_sa_full, ix->d_pac, ix->d_end_pos, ix->d_end_chr, ix->d_chr_fwd, ix->d_ktab, ix->d_rank, ix->d_rank2, ix->d_rank2_c2};
    for (void *q : p) if (q) (void)hipFree(q);
    // (a caller that frees the index before its contexts — allowed: a context that is only freed afterwards touches no device memory of the index — leaves
    //  the host object to the last mcx_ctx_free, which still counts itself out of it)
    ix->d_bwt = ix->d_sa = ix->d_sa_full = ix->d_pac = ix->d_end_pos = ix->d_end_chr = ix->d_chr_fwd = ix->d_ktab = ix->d_rank = ix->d_rank2 = ix->d_rank2_c2 = nullptr;
    if (ix->n_ctx.load() > 0) { ix->orphan.store(true); return; }
    delete ix;
}
// gives back what an index holds above `full_sa` (2 -> 1: the pair records).  Contexts made before keep their view of the index: close them first.
extern "C" int mcx_index_trim(mcx_index *ix, int full_sa)
{
    if (!ix) return fail(MCX_ERR_ARG, "mcx_index_trim: null argument");
    if (full_sa < 1) return fail(MCX_ERR_UNSUPPORTED, "mcx_index_trim: only the pair records can be released (full_sa = 1)");
    if (full_sa >= 2 || !ix->d_rank2) return 0;
    // a context keeps pointers into what goes (the view it copies per pass, a batch under way): none may be alive
    if (ix->n_ctx.load() > 0) return fail(MCX_ERR_ARG, "mcx_index_trim: " + std::to_string(ix->n_ctx.load()) + " context(s) of this index are still open; free them first");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(ix->d_rank2); (void)hipFree(ix->d_rank2_c2);
    ix->d_rank2 = ix->d_rank2_c2 = nullptr;
    ix->view.rank2 = nullptr; ix->view.rank2_c2 = nullptr;
    ix->hbm_bytes -= ix->rank2_bytes; ix->rank2_bytes = 0;
    return 0;
}
extern "C" int64_t mcx_index_genome_size(const mcx_index *ix) { return ix->host.G; }
extern "C" int32_t mcx_index_n_chr(const mcx_index *ix) { return (int32_t)ix->host.chr_len.size(); }
extern "C" const char *mcx_index_chr_name(const mcx_index *ix, int32_t i) { return ix->host.chr_name[i].c_str(); }
extern "C" int32_t mcx_index_chr_len(const mcx_index *ix, int32_t i) { return ix->host.chr_len[i]; }
extern "C" int64_t mcx_index_hbm_bytes(const mcx_index *ix) { return ix->hbm_bytes; }

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
struct PairSel {             // which pairs a launch works on
    const uint32_t *ids;     // batch pair id per local index (null: identity)
    const int32_t *est;      // EstiDistance per local index
    uint32_t n;
};

static __device__ __forceinline__ uint32_t sel_pair(const PairSel &s, uint32_t local) { return s.ids ? s.ids[local] : local; }

// The chromosome tables (PosChrIdMap as sorted arrays) are binary-searched several times per
// pair; each probe is a dependent load.  Blocks copy them to LDS once (when they fit) and the
// per-pair code then searches LDS through the same pointers.
struct EndsLds { int64_t pos[kLdsEnds]; int32_t chr[kLdsEnds]; };

static __device__ __forceinline__ void stage_ends(IndexView &ix, EndsLds &l)
{
    if (ix.n_ends <= kLdsEnds) { // uniform over the grid
        for (int i = threadIdx.x; i < ix.n_ends; i += blockDim.x) { l.pos[i] = ix.end_pos[i]; l.chr[i] = ix.end_chr[i]; }
        __syncthreads();
        ix.end_pos = l.pos; ix.end_chr = l.chr;
    }
}

// (a read without N also carries its 2-bit words — k_pack_reads' form, in HBM: a base then costs a shift of a word that
//  sits in L1 instead of a byte fetch and a table look-up, and gap fragments are compared sixteen bases at a time)
static __device__ __forceinline__ void make_reads(const Ctx &cx, const ReadBatch &rb, uint32_t pair, ReadRef rd[2])
{
    const int nr = cx.pm.paired ? 2 : 1;
    for (int s = 0; s < nr; s++) {
        uint32_t r = pair * nr + s;
        rd[s].ascii = rb.bases + rb.off[r];
        rd[s].rlen = (int32_t)(rb.off[r + 1] - rb.off[r]);
        rd[s].flipped = (cx.pm.paired && s == 1) ? 1 : 0;
        rd[s].codes = (cx.packed && !(cx.read_ext[r] >> 31)) ? cx.packed + (uint64_t)r * cx.wpad : nullptr;
#ifdef MCX_DEBUG_NO_CODES
        rd[s].codes = nullptr;
#endif
    }
}

struct SeedOut {
    uint2 *tasks;            // (local read, hit index) per hit to resolve
    uint32_t *n_tasks;
    uint32_t task_cap;
    uint32_t *read_ext;      // per batch read: extension steps, blocks touched
    uint32_t *read_blocks;
    const uint32_t *packed;  // 2-bit reads of the batch (k_pack_reads), wpad words each
    int wpad;
    uint32_t *queue;         // next read of the pass that no wavefront has taken yet
    // the late pairs' pass: their seed hits are already there, complete, in the records of the tier that listed them (pair = record:
    // the main pass) — a read without N takes them from there instead of searching again (null: every read is searched)
    const uint8_t *src_state; Layout src_lay; Caps src_caps;
};

// 16 bytes from any address: two aligned 16-byte fetches and a byte funnel
static __device__ __forceinline__ U4 load16_unaligned(const uint8_t *p)
{
    const uintptr_t a = (uintptr_t)p;
    const U4 *q = (const U4 *)(a & ~(uintptr_t)15);
    const U4 lo = q[0];
    const unsigned sh = (unsigned)(a & 15);
    if (sh == 0) return lo;
    const U4 hi = q[1];
    const unsigned qd = sh >> 2, b = sh & 3;
    uint32_t s0, s1, s2, s3, s4;
    switch (qd) {
    case 0: s0 = lo.x; s1 = lo.y; s2 = lo.z; s3 = lo.w; s4 = hi.x; break;
    case 1: s0 = lo.y; s1 = lo.z; s2 = lo.w; s3 = hi.x; s4 = hi.y; break;
    case 2: s0 = lo.z; s1 = lo.w; s2 = hi.x; s3 = hi.y; s4 = hi.z; break;
    default: s0 = lo.w; s1 = hi.x; s2 = hi.y; s3 = hi.z; s4 = hi.w; break;
    }
    U4 r;
    r.x = __builtin_amdgcn_alignbyte(s1, s0, b); r.y = __builtin_amdgcn_alignbyte(s2, s1, b);
    r.z = __builtin_amdgcn_alignbyte(s3, s2, b); r.w = __builtin_amdgcn_alignbyte(s4, s3, b);
    return r;
}

// four ASCII bytes (first base in the low byte) -> 8 code bits (first base on top) and 4 N flags
// (first base on top); `comp`: complement the bases (mate 2).  nt4_code (mcx_fm.h) on four lanes of
// one register: fold case, A/C/G/T membership by zero-byte tests, (c >> 1) & 3 hashes A0 C1 T2 G3.
// (lower: 4 flags likewise for bytes with bit 5 set — a lower-case letter where the byte is a base at all)
static __device__ __forceinline__ void pack4(uint32_t w, bool comp, uint32_t &codes, uint32_t &flags, uint32_t &lower)
{
    const uint32_t c = w & 0xDFDFDFDFu;
    auto zero_bytes = [](uint32_t t) { return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu); }; // 0x80 where a byte is 0
    const uint32_t member = (zero_bytes(c ^ 0x41414141u) | zero_bytes(c ^ 0x43434343u) | zero_bytes(c ^ 0x47474747u) | zero_bytes(c ^ 0x54545454u)) >> 7; // 0x01 per base
    const uint32_t h = (c >> 1) & 0x03030303u;
    uint32_t code = h ^ ((h >> 1) & 0x01010101u);
    if (comp) code ^= 0x03030303u;
    code &= member * 3u;
    const uint32_t y = __builtin_bswap32(code), n = __builtin_bswap32(member ^ 0x01010101u);
    codes = (y | (y >> 6) | (y >> 12) | (y >> 18)) & 0xFFu;
    flags = (n | (n >> 7) | (n >> 14) | (n >> 21)) & 0xFu;
    const uint32_t l = __builtin_bswap32((w >> 5) & 0x01010101u);
    lower = (l | (l >> 7) | (l >> 14) | (l >> 21)) & 0xFu;
}

// 16 oriented bases i0 .. i0+15 of a read -> (code word, MSB first; 16 N flags, bit 15 = base i0);
// bases past the read end give code 0 and flag 1; odd: 16 flags likewise for bytes of the read that are not an upper-case A C G T
// (what the -vcf bookkeeping asks of a read: mcx_profile.h), 0 past the read end
static __device__ __forceinline__ void pack16(const ReadRef &rd, int i0, uint32_t &codes, uint32_t &flags, uint32_t &odd)
{
    const int rlen = rd.rlen;
    int n = rlen - i0; if (n > 16) n = 16;
    codes = 0; flags = 0xFFFFu; odd = 0;
    if (n <= 0) return;
    // the span of the file's bytes under these bases: forward [i0, i0+n); mate 2 (reverse-complemented) [rlen-i0-n, rlen-i0) backwards
    const int a0 = rd.flipped ? rlen - i0 - n : i0;
    U4 v = load16_unaligned(rd.ascii + a0);
    if (rd.flipped) { // byte order reversed: the span's last byte first
        const U4 t = v;
        v.x = __builtin_bswap32(t.w); v.y = __builtin_bswap32(t.z); v.z = __builtin_bswap32(t.y); v.w = __builtin_bswap32(t.x);
    }
    uint32_t c0, f0, c1, f1, c2, f2, c3, f3, l0, l1, l2, l3;
    pack4(v.x, rd.flipped != 0, c0, f0, l0); pack4(v.y, rd.flipped != 0, c1, f1, l1); pack4(v.z, rd.flipped != 0, c2, f2, l2); pack4(v.w, rd.flipped != 0, c3, f3, l3);
    uint32_t c = (c0 << 24) | (c1 << 16) | (c2 << 8) | c3, f = (f0 << 12) | (f1 << 8) | (f2 << 4) | f3, o = f | (l0 << 12) | (l1 << 8) | (l2 << 4) | l3;
    if (n < 16) {
        const int pad = 16 - n;
        if (rd.flipped) { c <<= 2 * pad; f = (f << pad) & 0xFFFFu; o = (o << pad) & 0xFFFFu; } // the n bytes sat at the end of the reversed vector
        else { c &= ~0u << (2 * pad); o &= ~0u << pad; }
        f |= (1u << pad) - 1u;
        if (!rd.flipped) f &= 0xFFFFu;
    }
    codes = c; flags = f; odd = o & 0xFFFFu;
}

// The packed form of every read of a batch (mcx_fm.h pack_read: 16 bases per code word, one zero
// word, 32 N flags per mask word, one all-ones word).  One thread per 32 bases: two 16-byte spans
// of the file's bytes, fetched as aligned 16-byte words by neighbouring threads, so the batch is
// packed at streaming speed instead of base by base in the seeding lanes.  Mate 2 is packed
// reverse-complemented (ReadMapping.cpp:451).  tpr threads per read: ceil(max_read_len / 32) + 1.
// (any_n: set when a read of the batch holds a byte that is not one of ACGT — the passes of the large tier, which are queued after the
//  host has looked at the batch's counters anyway, leave out the kernel for such reads when there is none)
// (odd_flag, with the -vcf bookkeeping: bit 1 of the read's flag byte is set when the read holds a byte that is not an upper-case A C G T)
__global__ void __launch_bounds__(256) k_pack_reads(ReadBatch rb, int paired, int wpad, int tpr, uint32_t *out, uint32_t *any_n, uint8_t *odd_flag)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = (uint32_t)(t / (uint32_t)tpr);
    const int m = (int)(t % (uint32_t)tpr);
    if (r >= rb.n_reads) return;
    ReadRef rd;
    rd.ascii = rb.bases + rb.off[r]; rd.rlen = (int)(rb.off[r + 1] - rb.off[r]); rd.flipped = (paired && (r & 1)) ? 1 : 0;
    const int rlen = rd.rlen, nc = (rlen + 15) >> 4, nmw = (rlen + 31) >> 5;
    if (nc + 1 + nmw + 1 > wpad) return; // (a read longer than the context was sized for: the batch is refused before this runs, k_max_read_len)
    uint32_t *o = out + (uint64_t)r * wpad;
    if (m < nmw) {
        uint32_t c0, f0, c1, f1, o0, o1;
        pack16(rd, 32 * m, c0, f0, o0);
        pack16(rd, 32 * m + 16, c1, f1, o1);
        if (odd_flag && (o0 | o1)) atomicOr((uint32_t *)(odd_flag + (r & ~3u)), 2u << (8 * (r & 3)));
        o[2 * m] = c0;
        if (2 * m + 1 < nc) o[2 * m + 1] = c1;
        o[nc + 1 + m] = (f0 << 16) | f1;
        const int left = rlen - 32 * m; // bases of this mask word inside the read (bit 31 = the first)
        if (((f0 << 16) | f1) & (left >= 32 ? ~0u : ~(0xFFFFFFFFu >> left))) atomicOr(any_n, 1u);
    } else if (m == nmw) {
        o[nc] = 0u;
        o[nc + 1 + nmw] = 0xFFFFFFFFu;
    }
}

// Reads are handed to lanes as the lanes become free: a read is one to six searches (many more steps each inside a
// repeat), and a wave whose lanes each owned one read would run as long as its longest read while most lanes idle.
// The reads of a pass form one queue; a wavefront takes a chunk of it with one atomic whenever its lanes run dry, and a
// lane that has finished a read takes the chunk's next one — every search iteration of the wave finds (nearly) all
// lanes with work, and the launch ends when the queue does, not block by block.
constexpr int kSeedReadsPerLane = 4; // reads per lane and chunk (small selections: one, their launch is as long as its longest chain of reads)

// FM steps a lane takes before the wave looks at its other lanes again (MCX_SEED_FM_BUDGET for experiments)

static inline int seed_reads_per_lane(uint64_t n_reads) { return n_reads >= (uint64_t)1 << 21 ? kSeedReadsPerLane : (n_reads >= (uint64_t)1 << 19 ? 2 : 1); }

#ifndef MCX_SEED_WAVES
#define MCX_SEED_WAVES 1
#endif
#ifdef MCX_SEED_STATS
__device__ unsigned long long g_seed_hist[2][24]; // reads / index blocks by blocks per read (bucket = bit length of the count)
#endif
__global__ void __launch_bounds__(256, MCX_SEED_WAVES) k_seed(Ctx cx, ReadBatch rb, PairSel sel, SeedOut so, int pk_words, int reads_per_lane, int fm_budget)
{
    extern __shared__ uint32_t pk_lds[]; // packed reads: word k of lane t at pk_lds[k * blockDim.x + t]
    const int nr = cx.pm.paired ? 2 : 1;
    const uint32_t total = sel.n * nr;
    const int lane = threadIdx.x & 63;
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const uint32_t chunk = 64u * (uint32_t)reads_per_lane;
    PackedRead pk; pk.w = pk_lds + threadIdx.x; pk.stride = blockDim.x; pk.n_code = 0;
    bool have = false;
    uint32_t lr = 0, r = 0, nm = 0;
    int rlen = 0, p = 0, n = 0;
    int64_t ext = 0, blocks = 0;
    Hit *hits = nullptr;
    // the read is done: counters, and SA tasks for the hits that are still BWT rows (the others carry their text position)
    uint32_t has_n = 0;
    int cap = cx.caps.hit_seed;
    auto finish_read = [&]() {
        so.read_ext[r] = (uint32_t)ext | (has_n << 31); so.read_blocks[r] = (uint32_t)blocks | ((uint32_t)n << 20); // (ext < 2^31; blocks < 2^20; n < 2^12)
#ifdef MCX_SEED_STATS
        { const int bk = blocks ? 64 - __clzll((unsigned long long)blocks) : 0; atomicAdd(&g_seed_hist[0][bk], 1ull); atomicAdd(&g_seed_hist[1][bk], (unsigned long long)blocks); }
#endif
        if (cx.ix.sa_full) return; // every hit already carries its text position (seed_search)
        const int keep = n <= cx.caps.hit_seed ? n : 0; // overflowing reads are re-run in the next tier
        int todo = 0;
        for (int i = 0; i < keep; i++) if (!(hits[i].len & kHitResolved)) todo++;
        uint32_t at = todo ? atomicAdd(so.n_tasks, (uint32_t)todo) : 0u;
        for (int i = 0; i < keep; i++) {
            if (hits[i].len & kHitResolved) { hits[i].len &= ~kHitResolved; continue; }
            if (at < so.task_cap) so.tasks[at] = make_uint2(lr, (uint32_t)i);
            at++;
        }
    };
    SeedWalk walk; walk.phase = 0;
    uint32_t q_next = 0, q_end = 0; // the wave's chunk of the queue (the same in every lane)
    bool dry = false;               // the queue has nothing left
    for (;;) {
        // ---- lanes without a read take the next ones of the wave's chunk (the whole wave passes here every iteration) ----
        const uint64_t need = __ballot(!have);
        bool fresh = false;
        if (need && !dry) {
            const uint32_t rank = (uint32_t)__popcll(need & lt_mask), want = (uint32_t)__popcll(need);
            uint32_t given = 0;
            while (given < want) {
                if (q_next == q_end) {
                    uint32_t b = 0;
                    if (lane == 0) b = atomicAdd(so.queue, chunk);
                    b = __shfl(b, 0, 64);
                    if (b >= total) { dry = true; break; }
                    q_next = b; q_end = min(b + chunk, total);
                }
                const uint32_t take = min(want - given, q_end - q_next);
                if (!have && !fresh && rank >= given && rank < given + take) { lr = q_next + (rank - given); fresh = true; }
                q_next += take; given += take;
            }
        }
        if (fresh) {
            r = sel_pair(sel, lr / nr) * nr + lr % nr;
            rlen = (int)(rb.off[r + 1] - rb.off[r]);
            hits = pair_state(cx.state, cx.lay, cx.caps, lr / nr).hits[lr % nr];
            n = 0; p = 0; ext = 0; blocks = 0; has_n = 0; walk.phase = 0;
            const int words = packed_words(rlen);
            // (only when neither read of the pair holds an N: such a pair went through k_rescue in tier 0, whose rescue_mate appends seeds to the
            //  N-free read's hits — the same test as k_rescue_plan's)
            if (so.src_state && !((so.read_ext[r] | (nr == 2 ? so.read_ext[r ^ 1u] : 0u)) >> 31)) {
                // (the list as k_cluster left it: PosDiff > 0 only, sorted — clustering it again gives the same candidates; the entries the
                //  filter dropped are made up by entries it drops again, so that the read's hit count, a statistic, stays what the search found)
                const PairState src = pair_state((uint8_t *)so.src_state, so.src_lay, so.src_caps, r / nr);
                const int m = src.hdr->n_hits[r % nr], n0 = (int)(so.read_blocks[r] >> 20);
                for (int i = 0; i < m; i++) hits[i] = src.hits[r % nr][i];
                Hit none; none.gPos = 0; none.rPos = 0; none.len = 0;
                for (int i = m; i < n0; i++) hits[i] = none;
                fresh = false; // (counters untouched: they are the search's)
            } else
            if (rlen > 0 && words <= pk_words) {
                const U4 *src = (const U4 *)(so.packed + (uint64_t)r * so.wpad);
                for (int k = 0; k < words; k += 4) {
                    const U4 v = src[k >> 2];
                    pk.w[k * pk.stride] = v.x;
                    if (k + 1 < words) pk.w[(k + 1) * pk.stride] = v.y;
                    if (k + 2 < words) pk.w[(k + 2) * pk.stride] = v.z;
                    if (k + 3 < words) pk.w[(k + 3) * pk.stride] = v.w;
                }
                pk.n_code = ((rlen + 15) >> 4) + 1;
                { // does the read hold an N?  (mask words: bit 31-s of word s/32; bases past the end are flagged too)
                    const int nmw = (rlen + 31) >> 5;
                    uint32_t any = 0;
                    for (int m = 0; m < nmw; m++) {
                        uint32_t w = pk.w[(pk.n_code + m) * pk.stride];
                        if (m == nmw - 1 && (rlen & 31)) w &= ~(0xFFFFFFFFu >> (rlen & 31));
                        any |= w;
                    }
                    has_n = any ? 1u : 0u;
                }
                have = seed_next_start(pk, rlen, p, nm);
            }
            if (!have && fresh) finish_read(); // (nothing to search in it: the lane takes another one next time round)
        }
        if (!__ballot(have)) { if (dry) break; continue; }
        // ---- every lane that holds a read moves its search on: each phase with a budget, so that a lane deep inside a repeat
        //      (dozens of FM steps) holds the wave up for a few steps at a time while the others finish searches and take new reads ----
        if (have && walk.phase == 0) seed_begin(cx.ix, pk, rlen, nm, p, walk);
        if (have && walk.phase == 1) seed_fm(cx.ix, pk, rlen, p, walk, blocks, fm_budget);
        if (have && walk.phase == 2) seed_compare_wide(cx.ix, pk, rlen, p, walk, 1 << 30);
        if (have && walk.phase == 3) {
            seed_take(cx.ix, p, walk, hits, cap, n, ext);
            if (!seed_next_start(pk, rlen, p, nm)) { finish_read(); have = false; }
        }
    }
}

__global__ void __launch_bounds__(256) k_sa(Ctx cx, SeedOut so, int paired, uint32_t *lf_total)
{
    const uint32_t n = min(*so.n_tasks, so.task_cap);
    const int nr = paired ? 2 : 1;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const uint2 task = so.tasks[t];
        PairState st = pair_state(cx.state, cx.lay, cx.caps, task.x / nr);
        Hit &h = st.hits[task.x % nr][task.y];
        int lf = 0;
        h.gPos = (int64_t)fm_sa(cx.ix, (uint64_t)h.gPos, lf);
        if (lf_total && lf) atomicAdd(lf_total, (uint32_t)lf);
    }
}

struct RescueList { uint32_t *ids; uint32_t *n; uint32_t cap; };
// pairs that ran over the tier-0 capacities while clustering are listed right there, with their estimate: the large tier maps
// them on a stream of its own while the rest of the pass is still under way (ids null: no such list)
struct EarlyList { uint32_t *ids; int32_t *est; uint32_t *n; uint32_t cap; uint32_t *n_hits; };

// The per-pair kernels give every lane one pair, and a wavefront is as slow as its heaviest lane: next to a pair from a repeat
// (dozens of hits to sort and cluster, a dozen candidates to build and score) sixty-three ordinary pairs wait.  So the pairs
// of a pass are dealt to the lanes by weight — total seed hits, known once k_seed is done — heaviest class first: like sits
// with like, and the long wavefronts start early.  Two small passes (count, place) make the permutation `order`.
// (counters that many wavefronts add to sit one per 256 bytes: their atomics then run in different L2 channels instead of queueing on one line)
constexpr int kCntPad = 64;
constexpr int kWorkClasses = 6;
static __device__ __forceinline__ int work_class(uint32_t hits) { return hits > 64 ? 0 : hits > 32 ? 1 : hits > 16 ? 2 : hits > 8 ? 3 : hits > 4 ? 4 : 5; }

static __device__ __forceinline__ int pair_work_class(const PairSel &sel, uint32_t local, const uint32_t *read_blocks, int nr, const uint8_t *done = nullptr)
{
    if (done && done[local]) return -1; // k_simple took the pair from its seeds to its records: no lane of the per-pair kernels for it
    const uint32_t pair = sel_pair(sel, local);
    return work_class((read_blocks[pair * nr] >> 20) + (nr == 2 ? read_blocks[pair * nr + 1] >> 20 : 0u));
}

constexpr int kOrderTile = 16; // pairs per thread of the two passes: a block of 256 settles 4096 pairs with one global atomic per class

__global__ void __launch_bounds__(256) k_order_count(PairSel sel, const uint32_t *read_blocks, int nr, uint32_t *counts, const uint8_t *done)
{
    __shared__ uint32_t n_cls[kWorkClasses];
    if (threadIdx.x < kWorkClasses) n_cls[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * (256u * kOrderTile);
    for (int t = 0; t < kOrderTile; t++) {
        const uint32_t local = base + t * 256u + threadIdx.x;
        const int cls = local < sel.n ? pair_work_class(sel, local, read_blocks, nr, done) : -1;
#pragma unroll
        for (int k = 0; k < kWorkClasses; k++) {
            const uint64_t m = __ballot(cls == k);
            if ((threadIdx.x & 63) == 0 && m) atomicAdd(&n_cls[k], (uint32_t)__popcll(m));
        }
    }
    __syncthreads();
    if (threadIdx.x < kWorkClasses && n_cls[threadIdx.x]) atomicAdd(counts + threadIdx.x * kCntPad, n_cls[threadIdx.x]);
}

// counts[k * kCntPad]: pairs of class k; counts[(8 + k) * kCntPad]: how many of them were placed so far
__global__ void __launch_bounds__(256) k_order_place(PairSel sel, const uint32_t *read_blocks, int nr, uint32_t *counts, uint32_t *order, const uint8_t *done)
{
    __shared__ uint32_t n_cls[kWorkClasses], at_cls[kWorkClasses];
    if (threadIdx.x < kWorkClasses) n_cls[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * (256u * kOrderTile);
    const int lane = threadIdx.x & 63;
    int cls[kOrderTile];
#pragma unroll
    for (int t = 0; t < kOrderTile; t++) {
        const uint32_t local = base + t * 256u + threadIdx.x;
        cls[t] = local < sel.n ? pair_work_class(sel, local, read_blocks, nr, done) : -1;
#pragma unroll
        for (int k = 0; k < kWorkClasses; k++) {
            const uint64_t m = __ballot(cls[t] == k);
            if (lane == 0 && m) atomicAdd(&n_cls[k], (uint32_t)__popcll(m));
        }
    }
    __syncthreads();
    if (threadIdx.x < kWorkClasses) { // the block's stretch of every class: where the class begins + what other blocks took before
        uint32_t first = 0;
        for (int k = 0; k < (int)threadIdx.x; k++) first += counts[k * kCntPad];
        const uint32_t mine = n_cls[threadIdx.x];
        at_cls[threadIdx.x] = first + (mine ? atomicAdd(counts + (8 + threadIdx.x) * kCntPad, mine) : 0u);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kOrderTile; t++) {
        const uint32_t local = base + t * 256u + threadIdx.x;
#pragma unroll
        for (int k = 0; k < kWorkClasses; k++) {
            const uint64_t m = __ballot(cls[t] == k);
            uint32_t b = 0;
            if (lane == 0 && m) b = atomicAdd(&at_cls[k], (uint32_t)__popcll(m));
            b = __shfl(b, 0, 64);
            if (cls[t] == k) order[b + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = local;
        }
    }
}

// The straight-line pairs of a pass, from their seeds to their records (mcx_simple.h): one pair per lane, in pair order (their
// work is even: no dealing by weight, and neighbouring lanes write neighbouring records).  What a lane needs — at most four
// seeds per read, the 2-bit words of its reads, a few words of the genome under the gaps — stays in registers; the CIGAR
// operations wait word-major in LDS until the wave has reserved its words of the pool with one atomic.  done[pair] tells the
// per-pair kernels behind this one which pairs are left to them (k_order_*).
// A pair whose only obstacle is a small gapped extension or two (mcx_simple.h SimpleJob) writes the problems down and waits
// (done = 2): k_simple_dp solves them one per lane, k_simple_rest makes the pass again with the results at hand.
struct SimpleLater {        // the pairs that wait, and their problems
    uint32_t *pairs;        // [cap]
    SimpleJob *jobs;        // [cap * kSimpleJobs]: waiting pair i's at i * kSimpleJobs
    SimpleRes *res;         // the same places
    uint32_t *job_list;     // [cap * kSimpleJobs]: the places that hold a problem
    uint32_t *n_pairs, *n_jobs;
    uint32_t cap;
};

// DETAIL (the -vcf bookkeeping is on): the pair's two detail records — what write_detail leaves for a read with one surviving candidate, and
// pair_stats' discordant-pair fields in read 1's — come from here too (mcx_simple.h SimpleDetail), so that the profile does not switch the path off.
template <bool NW, int MODE, bool DETAIL>
__global__ void __launch_bounds__(256) k_simple(Ctx cx, ReadBatch rb, uint32_t n_pairs, const int32_t *est, const uint32_t *read_blocks, AlnRec *recs, PairOut *pout,
                                                uint8_t *done, uint32_t *pool_over, uint32_t *n_done, SimpleLater sl)
{
    __shared__ EndsLds ends;
    __shared__ uint16_t cig_stage[256 * 2 * kSimpleRuns]; // operation k of thread t at [k * 256 + t] (16 bits: a run is at most 4095 long)
    __shared__ uint32_t job_stage[MODE == kDpCollect ? 3 * kSimpleJobs * 256 : 1]; // word w of thread t's problem k at [(3 * k + w) * 256 + t]
    stage_ends(cx.ix, ends);
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; // collect: the pair; replay: the waiting pair
    const int nr = cx.pm.paired ? 2 : 1;
    uint16_t *stage = cig_stage + threadIdx.x;
    const uint32_t n_slots = MODE == kDpReplay ? min(*sl.n_pairs, sl.cap) : n_pairs;
    bool ok = slot < n_slots;
    const uint32_t pair = MODE == kDpReplay ? (ok ? sl.pairs[slot] : 0u) : slot;
    SimpleRead sr[2];
    int rl[2] = {0, 0};
    typename std::conditional<DETAIL, SimpleDetail, SimpleNoDetail>::type det;
    SimpleDpIo io;
    io.mode = MODE == kDpCollect && !sl.pairs ? kDpNone : MODE;
    io.jobs = MODE == kDpCollect ? job_stage + threadIdx.x : nullptr; io.job_stride = 256;
    io.res = MODE == kDpReplay ? sl.res + (uint64_t)slot * kSimpleJobs : nullptr;
    io.n = 0; io.read = 0;
    bool later = false;
    if (ok) {
        const PairState st = pair_state(cx.state, cx.lay, cx.caps, pair);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (s < nr && ok) {
                const uint32_t r = pair * nr + s;
                const int nh = (int)(read_blocks[r] >> 20);
                rl[s] = (int)(rb.off[r + 1] - rb.off[r]);
                io.read = r;
                ok = nh >= 1 && nh <= kSimpleHits && !(cx.read_ext[r] >> 31);
                if (ok) {
                    if constexpr (DETAIL) { uint8_t *rec = cx.detail + (uint64_t)r * cx.dlay.stride; det.frags = (Frag *)(rec + sizeof(DetailHdr)); det.ops = rec + cx.dlay.off_ops; }
                    const int how = simple_read<NW, uint16_t>(cx.ix, cx.pm, rl[s], cx.packed + (uint64_t)r * cx.wpad, st.hits[s], nh, sr[s], stage + s * kSimpleRuns * 256, 256, io, det);
                    ok = how != kSimpleNo;
                    later = later || how == kSimpleLater;
                }
            }
        }
        if (ok && nr == 2) ok = simple_pair_ok(sr[0], sr[1], est[pair]);
    }
    if (MODE == kDpCollect) { // the pairs that wait: a place among them, their problems into it
        const bool wait = ok && later;
        const uint32_t at = wave_reserve(sl.n_pairs, wait ? 1u : 0u);
        const bool kept = wait && at < sl.cap;
        const uint32_t jat = wave_reserve(sl.n_jobs, kept ? (uint32_t)io.n : 0u);
        if (kept) {
            sl.pairs[at] = pair;
            for (int k = 0; k < io.n; k++) {
                const uint32_t *w = job_stage + (3 * k) * 256 + threadIdx.x;
                sl.jobs[(uint64_t)at * kSimpleJobs + k] = simple_job_unpack(w[0], w[256], w[512], pair * (uint32_t)nr, nr);
                sl.job_list[jat + k] = at * kSimpleJobs + (uint32_t)k;
            }
            done[pair] = 2;
        }
        if (wait) ok = false; // (not kept: the general path)
        later = kept;
    }
    const uint32_t want = ok ? (uint32_t)sr[0].n_cig + (nr == 2 ? (uint32_t)sr[1].n_cig : 0u) : 0u;
    const uint32_t at = wave_reserve(cx.cig_pool_n, want);
    const uint64_t took = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && took) atomicAdd(n_done, (uint32_t)__popcll(took));
    if (slot >= n_slots) return;
    if (ok && at + want > cx.cig_pool_cap) { atomicOr(pool_over, 1u); ok = false; } // (the batch fails; the general path reports it)
    if (!later) done[pair] = ok ? 1 : 0;
    if (!ok) return;
    const uint32_t off[2] = {at, at + (uint32_t)sr[0].n_cig};
#pragma unroll
    for (int s = 0; s < 2; s++) {
        if (s >= nr) break;
        for (int k = 0; k < sr[s].n_cig; k++) cx.cig_pool[off[s] + k] = (uint32_t)stage[(s * kSimpleRuns + k) * 256];
    }
    AlnRec rec2[2];
    PairOut po;
    simple_pair(cx, nr == 2, sr[0], sr[nr - 1], rl[0], rl[nr - 1], est[pair], rec2, off, po);
    recs[(uint64_t)pair * nr] = rec2[0];
    if (nr == 2) recs[(uint64_t)pair * nr + 1] = rec2[1];
    pout[pair] = po;
    if constexpr (DETAIL) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (s >= nr) break;
            const DetailHdr d = simple_detail_hdr(cx.ix, nr == 2, s, sr[0], sr[nr - 1]);
            *(DetailHdr *)(cx.detail + ((uint64_t)pair * nr + s) * cx.dlay.stride) = d;
        }
    }
}

// the problems the straight-line pairs wrote down, one per lane (mcx_simple.h simple_dp_job): a strip of 16 columns, the lane's
// traceback words word-major in LDS
template <bool NW>
__global__ void __launch_bounds__(256) k_simple_dp(Ctx cx, SimpleLater sl)
{
    constexpr int W = 2 + kSimpleDp * LaneDir<16, NW>::words;
    __shared__ uint32_t words[W * 256];
    const uint32_t n = min(*sl.n_jobs, sl.cap * kSimpleJobs);
    LaneMem mem; mem.base = words; mem.stride = 256; mem.lane = threadIdx.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { // (the list's length stays on the device)
        const uint32_t at = sl.job_list[i];
        const SimpleJob j = sl.jobs[at];
        sl.res[at] = simple_dp_job<NW>(cx.ix, j, cx.packed + (uint64_t)j.read * cx.wpad, mem);
    }
}

// pairs the per-pair kernels work on: all of the selection, or — with k_simple ahead of them — those the order lists (its class counts)
static __device__ __forceinline__ uint32_t listed_pairs(const PairSel &sel, const uint32_t *order_cnt)
{
    if (!order_cnt) return sel.n;
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < kWorkClasses; k++) n += order_cnt[k * kCntPad];
    return n;
}

// (cls_lo .. cls_hi: with the pairs listed by weight, the launch takes the pairs of these classes only — the heavy classes go first, in
//  a launch of their own, because every pair that can run over this tier's capacities is among them: the large tier starts on
//  them while the light pairs, nine in ten, are still being clustered)
__global__ void __launch_bounds__(256) k_cluster(Ctx cx, ReadBatch rb, PairSel sel, RescueList rl, const uint32_t *read_blocks, EarlyList el, const uint32_t *order,
                                                 const uint32_t *order_cnt, int cls_lo, int cls_hi)
{
    __shared__ EndsLds ends;
    uint32_t first = 0, n_listed = listed_pairs(sel, order_cnt);
    if (order_cnt && order) {
        uint32_t at = 0;
        for (int k = 0; k < kWorkClasses; k++) { if (k == cls_lo) first = at; at += order_cnt[k * kCntPad]; if (k == cls_hi) n_listed = at; }
    }
    if (first + blockIdx.x * blockDim.x >= n_listed) return; // (uniform over the block)
    stage_ends(cx.ix, ends);
    const uint32_t slot = first + blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = slot < n_listed;
    const uint32_t local = in ? (order ? order[slot] : slot) : 0u;
    uint32_t need = 0, over = 0;
    if (in) {
        ReadRef rd[2];
        make_reads(cx, rb, sel_pair(sel, local), rd);
        PairState st = pair_state(cx.state, cx.lay, cx.caps, local);
        const uint32_t pair = sel_pair(sel, local);
        const int nr = cx.pm.paired ? 2 : 1;
        int nh[2] = {(int)(read_blocks[pair * nr] >> 20), nr == 2 ? (int)(read_blocks[pair * nr + 1] >> 20) : 0}; // k_seed's hit counts
        stage_cluster_pair(cx, local, rd, sel.est[local], nh);
        const uint32_t fl = st.hdr->flags;
        need = (cx.pm.paired && !(fl & kOvAny) && st.hdr->n_paired == 0) ? 1u : 0u;
        over = (el.ids && (fl & kOvAny)) ? 1u : 0u;
        if (over && (fl & kOvHits)) atomicAdd(el.n_hits, 1u); // (diagnosis: how many of the listed pairs were known to run over once k_seed was done)
        if (over) st.hdr->flags = fl | kDispatched; // the later stages of this tier leave the pair alone (they skip kOvAny) and k_finish writes nothing for it
        else if (need) st.hdr->flags = fl | kAwaitRescue;
    }
    const uint32_t at = wave_reserve(rl.n, need);
    if (need && at < rl.cap) rl.ids[at] = local;
    if (el.ids) {
        const uint32_t ea = wave_reserve(el.n, over);
        if (over && ea < el.cap) { el.ids[ea] = sel_pair(sel, local); el.est[ea] = sel.est[local]; }
    }
}

// behind k_cluster on its stream: the list's length — and whether the batch holds a read with an N — into page-locked host memory, so that the host,
// which drives the large tier for the listed pairs, needs no copy of its own behind the kernel (such a copy is a blit kernel that waits for a free CU on a chip
// the pass is filling: 0.3 to 0.9 ms before the large tier's first kernel could start).  (Counting the clustering kernel's workgroups as they end, so
// that the last one could write the words, cost the kernel 0.5 ms: 15 000 atomics on one address.)
__global__ void k_publish_early(const uint32_t *n, const uint32_t *any_n, volatile uint32_t *host)
{
    if (threadIdx.x == 0) { host[0] = *n; host[1] = *any_n; __threadfence_system(); }
}

// ---- clustering and pairing of a heavy pair by a whole wavefront -----------------------------------------------------------
// The large tier's pairs come from repeats: hundreds of seed hits per read, hundreds of candidates.  One lane that sorts 800
// hits and pairs 300 x 300 candidates in its pair record (every step a dependent fetch from HBM) takes milliseconds, and the
// launch is as long as its slowest lane.  Here a wavefront takes a pair: a read's hits are brought into LDS, sorted there by all
// 64 lanes (bitonic, keys (PosDiff, rPos)), clustered by one lane out of LDS (the scan is serial by nature — its threshold
// moves with every cluster — but a step is an LDS access now, not a trip to HBM), and the candidates of the two reads are paired
// with the candidates of read 1 spread over the lanes.  Same functions, same results as stage_cluster_pair.
struct ClusterLds { // per-wave arrays in dynamic LDS (cap = the tier's candidate capacity per read)
    Hit *hits;          // [hit_cap]: the read being clustered
    int64_t *pd[2];     // [cand_cap] per read: PosDiff of the candidate's first seed
    int32_t *score[2];  // score (0: dropped)
    uint32_t *span[2];  // first seed | seeds << 16
    int32_t *mate[2];   // PairedAlnCanIdx
    int32_t *pick;      // [cand_cap]: the partner a candidate of read 1 chose (pairing scratch)
};

static inline MCX_HD int cluster_sort_room(int hit_cap) { int p = 1; while (p < hit_cap) p <<= 1; return p; } // (the sort works on a power of two)
static inline size_t cluster_lds_bytes(int hit_cap, int cand_cap) { return (size_t)cluster_sort_room(hit_cap) * sizeof(Hit) + (size_t)cand_cap * (2 * (8 + 4 + 4 + 4) + 4) + 64; }
constexpr int kClusterSmall = 192; // hits per read up to which a pair takes the launch with the small share of LDS (more wavefronts per CU)

// (hits_lo < hits <= hits_hi: the pairs of this launch, by the larger hit count of their reads; lds_hits / lds_cands: what its LDS holds)
__global__ void __launch_bounds__(64) k_cluster_wave(Ctx cx, ReadBatch rb, PairSel sel, RescueList rl, const uint32_t *read_blocks, int hits_lo, int hits_hi, int lds_hits,
                                                     int lds_cands)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t cl_lds[];
    __shared__ EndsLds ends;
    stage_ends(cx.ix, ends);
    const int lane = threadIdx.x;
    const int nr = cx.pm.paired ? 2 : 1;
    ClusterLds L;
    {
        uint8_t *p = cl_lds;
        L.hits = (Hit *)p; p += (size_t)cluster_sort_room(lds_hits) * sizeof(Hit);
        for (int s = 0; s < 2; s++) { L.pd[s] = (int64_t *)p; p += (size_t)lds_cands * 8; }
        for (int s = 0; s < 2; s++) { L.score[s] = (int32_t *)p; p += (size_t)lds_cands * 4; }
        for (int s = 0; s < 2; s++) { L.span[s] = (uint32_t *)p; p += (size_t)lds_cands * 4; }
        for (int s = 0; s < 2; s++) { L.mate[s] = (int32_t *)p; p += (size_t)lds_cands * 4; }
        L.pick = (int32_t *)p;
    }
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_s_barrier(); };
    for (uint32_t local = blockIdx.x; local < sel.n; local += gridDim.x) {
        const uint32_t pair = sel_pair(sel, local);
        {
            const int h0 = (int)(read_blocks[pair * nr] >> 20), h1 = nr == 2 ? (int)(read_blocks[pair * nr + 1] >> 20) : 0, hm = h0 > h1 ? h0 : h1;
            if (hm <= hits_lo || hm > hits_hi) continue; // another launch's pair
        }
        PairState st = pair_state(cx.state, cx.lay, cx.caps, local);
        uint32_t flags = 0;
        int n_hits[2] = {0, 0}, n_cands[2] = {0, 0};
        for (int s = 0; s < nr; s++) {
            const uint32_t r = pair * nr + s;
            const int rlen = (int)(rb.off[r + 1] - rb.off[r]);
            int nh = (int)(read_blocks[r] >> 20);
            if (nh > cx.caps.hit_seed) { flags |= kOvHits; nh = 0; }
            Hit *g_hits = st.hits[s];
            // the read's hits with PosDiff > 0 into LDS, closed up (IdentifySimplePairs' tail, ReadMapping.cpp:141-152)
            wave_sync();
            int m = 0;
            for (int base = 0; base < nh; base += 64) {
                const int i = base + lane;
                Hit x; x.gPos = 0; x.rPos = 0; x.len = 0;
                bool keep = false;
                if (i < nh) { x = g_hits[i]; keep = hit_pd(x) > 0; }
                const uint64_t mask = __ballot(keep);
                if (keep) L.hits[m + __popcll(mask & ((1ull << lane) - 1ull))] = x;
                m += __popcll(mask);
            }
            int P = 1;
            while (P < m) P <<= 1;
            for (int i = m + lane; i < P; i += 64) { Hit x; x.gPos = (int64_t)1 << 62; x.rPos = 0; x.len = 0; L.hits[i] = x; } // (sort to the end)
            wave_sync();
            for (int k = 2; k <= P; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = lane; i < P; i += 64) {
                        const int o = i ^ j;
                        if (o > i) {
                            const Hit a = L.hits[i], b = L.hits[o];
                            const int64_t pa = hit_pd(a), pb = hit_pd(b);
                            const bool a_gt = pa > pb || (pa == pb && a.rPos > b.rPos);
                            if (((i & k) == 0) == a_gt) { L.hits[i] = b; L.hits[o] = a; }
                        }
                    }
                    wave_sync();
                }
            for (int i = lane; i < m; i += 64) g_hits[i] = L.hits[i]; // the later stages read the seeds in this order
            // clusters: a serial scan (its threshold moves with every cluster it keeps), out of LDS
            int nc = 0;
            if (lane == 0)
                nc = cluster_seeds_to(cx.ix, cx.pm, rlen, L.hits, m, [&](int k, int score, int first, int count, int64_t pd0) {
                    if (k < cx.caps.cand_seed) { L.pd[s][k] = pd0; L.score[s][k] = score; L.span[s][k] = (uint32_t)first | ((uint32_t)count << 16); }
                });
            nc = __shfl(nc, 0, 64);
            if (nc > cx.caps.cand_seed) { flags |= kOvCands; nc = 0; }
            n_hits[s] = m; n_cands[s] = nc;
            wave_sync();
        }
        for (int s = 0; s < 2; s++) for (int i = lane; i < n_cands[s]; i += 64) L.mate[s][i] = -1;
        wave_sync();
        // CheckPairedAlignmentDistance (pair_by_distance): read 1's candidates over the lanes
        int n_paired = 0, lo = 0, hi = 0x7fffffff;
        if (cx.pm.paired && !(flags & kOvAny)) {
            const int n1 = n_cands[0], n2 = n_cands[1];
            const int64_t est = (int64_t)sel.est[local];
            if (n1 * n2 > 100) // RemoveRedundantAlnCan on both
                for (int s = 0; s < 2; s++) {
                    int best = 0;
                    for (int i = lane; i < n_cands[s]; i += 64) best = max(best, L.score[s][i]);
                    for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o, 64));
                    if (n_cands[s] > 1) for (int i = lane; i < n_cands[s]; i += 64) if (L.score[s][i] < best) L.score[s][i] = 0;
                }
            wave_sync();
            int64_t max_lt = -1, min_ge = 0x7fffffff, top = 0;
            for (int i = lane; i < n1; i += 64) {
                const int sa = L.score[0][i];
                const int64_t pa = L.pd[0][i];
                int pick = -1, ps = 0;
                if (sa != 0)
                    for (int j = 0; j < n2; j++) {
                        const int sj = L.score[1][j];
                        const int64_t pj = L.pd[1][j];
                        if (sj == 0 || pj < pa) continue;
                        const int64_t d = pj - pa;
                        if (d < est) { if (d > max_lt) max_lt = d; if (sj > ps) { pick = j; ps = sj; } }
                        else if (d < min_ge) min_ge = d;
                    }
                L.pick[i] = pick;
                if (pick >= 0 && (int64_t)sa + ps > top) top = (int64_t)sa + ps;
            }
            for (int o = 32; o > 0; o >>= 1) {
                const int64_t a = __shfl_xor(max_lt, o, 64), b = __shfl_xor(min_ge, o, 64), c = __shfl_xor(top, o, 64);
                if (a > max_lt) max_lt = a;
                if (b < min_ge) min_ge = b;
                if (c > top) top = c;
            }
            wave_sync();
            if (top > 0)
                for (int i = lane; i < n1; i += 64) {
                    const int pick = L.pick[i];
                    if (pick >= 0 && (int64_t)L.score[0][i] + L.score[1][pick] == top) {
                        n_paired++;
                        L.mate[0][i] = pick;
                        atomicMax(&L.mate[1][pick], i); // (read 1's candidates are gone through in order: the last one that picks it stays)
                    }
                }
            for (int o = 32; o > 0; o >>= 1) n_paired += __shfl_xor(n_paired, o, 64);
            lo = (int)(max_lt + 1); hi = (int)min_ge;
            wave_sync();
        }
        // the candidates as the later stages find them in the pair record
        for (int s = 0; s < nr; s++)
            for (int i = lane; i < n_cands[s]; i += 64) {
                Cand c;
                c.score = L.score[s][i]; c.mate = (int16_t)L.mate[s][i]; c.first = (int16_t)(L.span[s][i] & 0xFFFFu); c.count = (int16_t)(L.span[s][i] >> 16); c.pd0 = L.pd[s][i];
                c.frag_off = 0; c.n_frags = 0; c.flag = 0; c.fwd = 1; c.in_pool = 0; c.pad = 0; c.pool_off = 0;
                st.cands[s][i] = c;
            }
        if (lane == 0) {
            PairHdr h;
            h.flags = flags; h.n_frags = 0; h.n_ops = 0; h.pair_dist = 0; h.n_jobs = 0; h.pair_ok = 0; h.mapped = 0; h.pad[0] = h.pad[1] = 0;
            h.n_hits[0] = (int16_t)n_hits[0]; h.n_hits[1] = (int16_t)n_hits[1]; h.n_cands[0] = (int16_t)n_cands[0]; h.n_cands[1] = (int16_t)n_cands[1];
            h.sum[0].best = h.sum[1].best = -1; h.sum[0].score = h.sum[1].score = 0; h.sum[0].sub = h.sum[1].sub = 0;
            h.est = sel.est[local]; h.est_lo = lo; h.est_hi = hi; h.n_paired = (int16_t)n_paired;
            if (cx.pm.paired && !(flags & kOvAny) && n_paired == 0) h.flags |= kAwaitRescue;
            *st.hdr = h;
            if (cx.pm.paired && !(flags & kOvAny) && n_paired == 0) {
                const uint32_t at = atomicAdd(rl.n, 1u);
                if (at < rl.cap) rl.ids[at] = local;
            }
        }
    }
}

constexpr int kRescueThreads = 256;
constexpr unsigned kRescueBlocks = 4096;
// HBM scratch per workgroup for reads with N (RescueWave::window_ids): the read's ids, the longest window's ids and bytes
constexpr size_t kRescueScratchWords = 1024 + (4096 + 64 + 8) + (4096 + 64) / 16 + 8;

// KG: the longest window the tier evaluates (caps.kmer_cap)
template <int KG>
#ifndef MCX_RESCUE_WAVES
#define MCX_RESCUE_WAVES 1
#endif
__global__ void __launch_bounds__(kRescueThreads, MCX_RESCUE_WAVES) k_rescue(Ctx cx, ReadBatch rb, PairSel sel, RescueList rl, uint32_t *kscratch)
{
    // one workgroup per unpaired pair (they are few, and one pair's windows are a long serial chain
    // for a single wavefront); read and window live in LDS as bit planes (RescueWave)
    __shared__ uint32_t q[3 * kRescueQWords];
    __shared__ uint32_t w[2 * rescue_wwords(KG)];
    __shared__ uint32_t ew[kRescueQWords];
    __shared__ int red[2 * (kRescueThreads / 64) + 5];
    const uint32_t n = min(*rl.n, rl.cap);
    RescueWave ev; ev.q = q; ev.w = w; ev.ew = ew; ev.wstride = rescue_wwords(KG); ev.red = red;
    ev.kq = kscratch + (size_t)blockIdx.x * kRescueScratchWords; ev.kg = ev.kq + 1024;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t local = rl.ids[i];
        ReadRef rd[2];
        make_reads(cx, rb, sel_pair(sel, local), rd);
        stage_rescue(cx, local, rd, ev);
        __syncthreads();
    }
}

// ---- mate rescue, window by window ---------------------------------------------------------------------------------------
// k_rescue gives a pair to one workgroup, which evaluates the pair's windows one after the other: a launch is as long as the
// pair with the most candidates to try (hundreds in the large tier) and its barriers.  But a window's evaluation depends on
// nothing but the candidate it lies beside — only *taking* the results is ordered (the mate's candidate and seed lists grow in
// window order, AlignmentRescue.cpp:28-111).  So: k_rescue_plan lists the windows of every listed pair (one lane per pair: the
// tests of rescue_mate up to the evaluation), k_rescue_eval evaluates the windows, one wavefront each, on bit planes in its
// share of LDS, and k_rescue_apply takes the results pair by pair in window order (one lane per pair: the rest of rescue_mate).
// Pairs with a read that holds an N keep k_rescue (the reference's 8-mer ids fall out of step after an N: RescueWave::window_ids).
struct RescueTask {
    uint32_t local;      // the pair (index in the pass)
    uint16_t side, ci;   // side 0: read 2 next to candidate ci of read 1; side 1: the other way round
    int32_t slen, sb;    // window length; the score a window has to beat
    int64_t left;
    int32_t pad[2];
};
static_assert(sizeof(RescueTask) == 32, "RescueTask is two 16-byte records");
struct RescueRes { int32_t score, d, n_seeds; uint32_t seed_off; }; // (seeds: n_seeds entries of the pass's seed pool from seed_off on)
struct RescuePlan { uint32_t local, first, n, flags; }; // a listed pair's windows [first, first + n) and the flags rescue_mate adds

struct RescueWork {
    RescueTask *tasks; RescueRes *res; Hit *seeds; RescuePlan *plans;
    uint32_t *n_tasks, *n_plans, *n_seeds;
    uint32_t task_cap, seed_cap; // windows the list holds; seeds the pool holds
    uint32_t *ids_n, *n_ids_n; // pairs left to k_rescue (a read with N)
};

__global__ void __launch_bounds__(256) k_rescue_plan(Ctx cx, ReadBatch rb, PairSel sel, RescueList rl, RescueWork rw)
{
    __shared__ EndsLds ends;
    stage_ends(cx.ix, ends);
    const IndexView &ix = cx.ix;
    const uint32_t n = min(*rl.n, rl.cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t local = rl.ids[i];
        const uint32_t pair = sel_pair(sel, local);
        PairState st = pair_state(cx.state, cx.lay, cx.caps, local);
        const PairHdr h = *st.hdr;
        if ((h.flags & kOvAny) || h.n_paired != 0) continue; // (stage_rescue's test)
        if ((cx.read_ext[2 * pair] | cx.read_ext[2 * pair + 1]) >> 31) { // a read with N: the id path
            const uint32_t at = atomicAdd(rw.n_ids_n, 1u);
            rw.ids_n[at] = local; // (as long as the list of listed pairs)
            continue;
        }
        const int rlen[2] = {(int)(rb.off[2 * pair + 1] - rb.off[2 * pair]), (int)(rb.off[2 * pair + 2] - rb.off[2 * pair + 1])};
        const int n1 = h.n_cands[0], n2 = h.n_cands[1];
        const Cand *c1 = st.cands[0], *c2 = st.cands[1];
        int s1 = 0, s2 = 0;
        for (int k = 0; k < n1; k++) { const int sc = c1[k].score; if (sc > s1) s1 = sc; }
        for (int k = 0; k < n2; k++) { const int sc = c2[k].score; if (sc > s2) s2 = sc; }
        int mode;
        if (s1 < (rlen[0] >> 2) && s2 < (rlen[1] >> 2)) continue; // rescue_mate returns 0 and leaves the pair alone
        else if (s1 - s2 > (rlen[1] >> 2)) mode = 1;
        else if (s2 - s1 > (rlen[0] >> 2)) mode = 2;
        else mode = 3;
        uint32_t flags = kRescueUsedEst;
        const int64_t est = (int64_t)(uint32_t)h.est;
        // two passes over the candidates: count the windows, reserve their places, write them
        uint32_t first = 0, count = 0;
        for (int pass = 0; pass < 2; pass++) {
            uint32_t at = first;
            for (int side = 0; side < 2; side++) {
                if (side == 0 && !(mode == 1 || mode == 3)) continue;
                if (side == 1 && !(mode == 2 || mode == 3)) continue;
                const int qlen = side == 0 ? rlen[1] : rlen[0];
                const Cand *ca = side == 0 ? c1 : c2;
                const int na = side == 0 ? n1 : n2, sa = side == 0 ? s1 : s2, sb = side == 0 ? s2 : s1;
                const int thr = sa >> 1;
                for (int ci = 0; ci < na; ci++) {
                    const Cand c = ca[ci];
                    if (c.score < thr || c.mate != -1) continue;
                    const int64_t left = side == 0 ? c.pd0 : c.pd0 - est;
                    int64_t right = side == 0 ? c.pd0 + est + qlen : c.pd0 + qlen;
                    if (right > ix.G2) right = ix.G2;
                    if (left < 0 || right >= ix.G2) continue;
                    const int e1 = end_slot(ix, left), e2 = end_slot(ix, right);
                    if (e1 < 0 || e2 < 0 || ix.end_chr[e1] != ix.end_chr[e2]) continue;
                    const int slen = (int)(right - left);
                    if (slen < qlen) continue;
                    if (slen > cx.caps.kmer_cap) { flags |= kOvKmer; continue; }
                    if (pass == 0) count++;
                    else {
                        if (at < rw.task_cap) {
                            RescueTask t; t.local = local; t.side = (uint16_t)side; t.ci = (uint16_t)ci; t.slen = slen; t.sb = sb; t.left = left; t.pad[0] = t.pad[1] = 0;
                            rw.tasks[at] = t;
                        }
                        at++;
                    }
                }
            }
            if (pass == 0) first = count ? atomicAdd(rw.n_tasks, count) : 0u;
        }
        RescuePlan pl; pl.local = local; pl.first = first; pl.n = count; pl.flags = flags;
        rw.plans[atomicAdd(rw.n_plans, 1u)] = pl; // (as long as the list of listed pairs)
    }
}

template <int KG>
__global__ void __launch_bounds__(256) k_rescue_eval(Ctx cx, ReadBatch rb, PairSel sel, RescueWork rw)
{
    constexpr int kWaves = 4, kW = rescue_wwords(KG);
    __shared__ uint32_t q_all[kWaves][3 * kRescueQWords];
    __shared__ uint32_t w_all[kWaves][2 * kW];
    __shared__ uint32_t ew_all[kWaves][kRescueQWords];
    __shared__ uint16_t ss_all[kWaves][128]; // where the seeds of a window's best diagonal begin (a read of 1024 bases has at most 94)
    const IndexView &ix = cx.ix;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    RescueWave ev; ev.q = q_all[wv]; ev.w = w_all[wv]; ev.ew = ew_all[wv]; ev.wstride = kW; ev.red = nullptr; ev.kq = ev.kg = nullptr;
    uint32_t *ql = ev.q, *qh = ev.q + kRescueQWords, *qv = ev.q + 2 * kRescueQWords, *wl = ev.w, *wh = ev.w + kW;
    const uint32_t n = min(*rw.n_tasks, rw.task_cap);
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    for (uint32_t t = blockIdx.x * kWaves + wv; t < n; t += gridDim.x * kWaves) {
        const RescueTask task = rw.tasks[t];
        const uint32_t pair = sel_pair(sel, task.local), r = 2 * pair + (task.side == 0 ? 1u : 0u);
        const int qlen = (int)(rb.off[r + 1] - rb.off[r]), slen = task.slen;
        const uint32_t *codes = cx.packed + (uint64_t)r * cx.wpad; // the read as mapped: mate 2 reverse-complemented (no N: k_rescue_plan)
        wave_sync();
        // the read's planes (base i at bit i & 31 of word i >> 5) and where a base counts (inside the read)
        for (int k = lane; k < kRescueQWords; k += 64) {
            uint32_t lo = 0, hi = 0, in = 0;
            if (32 * k < qlen) {
                const uint32_t a = codes[2 * k], b = 32 * k + 16 < qlen ? codes[2 * k + 1] : 0u;
                lo = (__brev(even_bits16(a)) >> 16) | (__brev(even_bits16(b)) & 0xFFFF0000u);
                hi = (__brev(even_bits16(a >> 1)) >> 16) | (__brev(even_bits16(b >> 1)) & 0xFFFF0000u);
                in = qlen - 32 * k >= 32 ? ~0u : ((1u << (qlen - 32 * k)) - 1u);
                lo &= in; hi &= in;
            }
            ql[k] = lo; qh[k] = hi; qv[k] = in;
        }
        wave_sync();
        uint32_t v8 = 0;
        if (lane < kRescueQWords) { // an 8-mer starts where eight bases of the read follow one another
            const uint32_t v = qv[lane], vn = lane + 1 < kRescueQWords ? qv[lane + 1] : 0u;
            v8 = v;
#pragma unroll
            for (int j = 1; j < kKmerSize; j++) v8 &= __funnelshift_r(v, vn, j);
        }
        wave_sync();
        if (lane < kRescueQWords) qv[lane] = v8;
        // the window's planes (RescueWave::window)
        const int pad = ((qlen + 31) & ~31) + 32;
        const int n_words = (pad + slen + qlen) / 32 + 3;
        for (int j = lane; j < n_words; j += 64) {
            uint32_t lo = 0, hi = 0;
            const int p0 = 32 * j - pad;
            if (p0 >= 0 && p0 < slen) {
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int p = p0 + 16 * half;
                    if (p >= slen) break;
                    const uint32_t x = ref_codes16(ix, task.left + p);
                    uint32_t l16 = __brev(even_bits16(x)) >> 16, h16 = __brev(even_bits16(x >> 1)) >> 16;
                    if (slen - p < 16) { const uint32_t keep = (1u << (slen - p)) - 1u; l16 &= keep; h16 &= keep; }
                    lo |= l16 << (16 * half); hi |= h16 << (16 * half);
                }
            }
            wl[j] = lo; wh[j] = hi;
        }
        wave_sync();
        const int d_lo = -(qlen - kKmerSize - 2), n_diag = slen - kKmerSize - 2 - d_lo + 1;
        uint32_t key = 0;
        for (int g = lane; g < n_diag; g += 64) {
            const int total = ev.diagonal(d_lo + g, qlen, slen, pad, false);
            const uint32_t k2 = ((uint32_t)total << 13) | (uint32_t)(8191 - g);
            if (total > 0 && k2 > key) key = k2;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)key, o, 64); if (other > key) key = other; }
        RescueRes res; res.score = (int)(key >> 13); res.d = 0; res.n_seeds = 0; res.seed_off = 0;
        if (key != 0) {
            res.d = d_lo + (8191 - (int)(key & 8191u));
            if (res.score > task.sb) { // (a window that does not beat the mate's best is dropped whatever its seeds)
                for (int k = lane; k < kRescueQWords; k += 64) ev.ew[k] = 0u;
                wave_sync();
                if (lane == 0) ev.diagonal(res.d, qlen, slen, pad, true); // the best diagonal's 8-mer match words, a word per 32 rows
                wave_sync();
                // The seeds of the diagonal: runs of three or more 8-mer matches (ten or more matching bases), in the order of their rows.  A lane per
                // word: where such a run begins and where it ends are two bit masks, the j-th beginning and the j-th end of the wavefront are one run's, and
                // two counts summed across the lanes number them.  (One lane walked the read's rows before, twice — counting, then writing —, a dependent
                // LDS read per row: some 7 000 instructions of a wavefront's time per window that beat its mate, more than the window's 1 300 diagonals cost.)
                uint32_t S = 0, E = 0;
                if (lane < kRescueQWords) {
                    const uint32_t e0 = ev.ew[lane], e1 = lane + 1 < kRescueQWords ? ev.ew[lane + 1] : 0u, ep = lane > 0 ? ev.ew[lane - 1] : 0u;
                    const uint32_t t = e0 & __funnelshift_r(e0, e1, 1) & __funnelshift_r(e0, e1, 2); // a triple of matches begins here
                    const uint32_t t_before = (ep >> 31) & e0 & (e0 >> 1) & 1u;                      // ... at the last row of the word before
                    const uint32_t t_after = e1 & (e1 >> 1) & (e1 >> 2) & 1u;                         // ... at the first row of the word after
                    S = t & ~((t << 1) | t_before);
                    E = t & ~((t >> 1) | (t_after << 31));
                }
                const int cs = __popc(S), ce = __popc(E);
                int is = cs, ie = ce;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int vs = __shfl_up(is, o, 64), ve = __shfl_up(ie, o, 64);
                    if (lane >= o) { is += vs; ie += ve; }
                }
                const int n_seeds = __shfl(is, 63, 64);
                uint32_t off = 0;
                if (lane == 0) off = atomicAdd(rw.n_seeds, (uint32_t)n_seeds);
                off = (uint32_t)__shfl((int)off, 0, 64);
                res.n_seeds = n_seeds; res.seed_off = off;
                if ((uint64_t)off + (uint64_t)n_seeds <= rw.seed_cap) { // (else the pool ran over: the pass is repeated in halves)
                    uint16_t *ss = ss_all[wv];
                    int j = is - cs;
                    for (uint32_t m = S; m; m &= m - 1u) ss[j++] = (uint16_t)(32 * lane + __ffs((int)m) - 1);
                    wave_sync();
                    Hit *out = rw.seeds + off;
                    j = ie - ce;
                    for (uint32_t m = E; m; m &= m - 1u, j++) {
                        const int start = ss[j], end = 32 * lane + __ffs((int)m) - 1; // the run's first and last triple: its 8-mer matches end two rows later
                        Hit h; h.rPos = start; h.gPos = (int64_t)(start + res.d) + task.left; h.len = end - start + 1 + kKmerSize + 1;
                        out[j] = h;
                    }
                }
            }
        }
        if (lane == 0) rw.res[t] = res;
    }
}

__global__ void __launch_bounds__(256) k_rescue_apply(Ctx cx, RescueWork rw)
{
    const uint32_t n = *rw.n_plans;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const RescuePlan pl = rw.plans[i];
        PairState st = pair_state(cx.state, cx.lay, cx.caps, pl.local);
        PairHdr h = *st.hdr;
        int nc[2] = {h.n_cands[0], h.n_cands[1]}, nh[2] = {h.n_hits[0], h.n_hits[1]};
        uint32_t flags = pl.flags;
        int paired = 0;
        for (uint32_t k = 0; k < pl.n; k++) {
            const uint32_t t = pl.first + k;
            if (t >= rw.task_cap) break; // (the list ran over: the pass is repeated in halves)
            const RescueRes res = rw.res[t];
            if (res.n_seeds == 0) continue;
            const RescueTask task = rw.tasks[t];
            if (res.score <= task.sb) continue;
            const int a = task.side, b = 1 - task.side; // candidate ci of read a gets a mate among read b's
            if ((uint64_t)res.seed_off + (uint64_t)res.n_seeds > rw.seed_cap) break; // (the pool ran over)
            if (nc[b] >= cx.caps.cand_cap) { flags |= kOvCands; continue; }
            paired++;
            // the new candidate's seeds stay in the pool (stage_build reads them there): no copy, and no bound on how many
            // seeds mate rescue may add to a read (a candidate per window, up to rlen / 11 seeds each)
            st.cands[a][task.ci].mate = (int16_t)nc[b];
            Cand nw;
            nw.score = res.score; nw.mate = (int16_t)task.ci; nw.first = 0; nw.count = (int16_t)res.n_seeds; nw.pd0 = (int64_t)res.d + task.left;
            nw.frag_off = 0; nw.n_frags = 0; nw.flag = 0; nw.fwd = 1; nw.in_pool = 1; nw.pad = 0; nw.pool_off = (int32_t)res.seed_off;
            st.cands[b][nc[b]] = nw;
            nc[b]++;
        }
        h.flags |= flags;
        h.n_cands[0] = (int16_t)nc[0]; h.n_cands[1] = (int16_t)nc[1];
        h.n_hits[0] = (int16_t)nh[0]; h.n_hits[1] = (int16_t)nh[1];
        h.n_paired = (int16_t)paired;
        *st.hdr = h;
    }
}

// fragment lists + DP problems of every pair; the problems are appended to one list per size
// class (mcx_glue.h dp_class) with one atomic per wave and class
// (late: the pairs that ran over this tier's capacities since clustering — mate rescue's additions, fragment lists, DP
//  columns, job lists — are listed like the early ones, for a second pass of the large tier beside the rest of this one)
#ifndef MCX_BUILD_WAVES
#define MCX_BUILD_WAVES 4 // (with the light pairs gone to k_simple what is left is heavier per lane: 128 registers, no spills, beat five waves at 96 with 51 spilled)
#endif
// (mode 0: every listed pair.  Mate rescue touches one pair in eighty, and the other seventy-nine need nothing from it: mode 1 builds
//  the pairs that do not await it — beside the rescue kernels, on another stream — and mode 2, once those are through, the pairs of
//  the rescue list rl.)
__global__ void __launch_bounds__(256, MCX_BUILD_WAVES) k_build(Ctx cx, ReadBatch rb, PairSel sel, JobSinks sinks, uint32_t *cells, uint32_t *unsupported, EarlyList late,
                                               const uint32_t *order, const uint32_t *order_cnt, int mode, RescueList rl)
{
    __shared__ EndsLds ends;
    const uint32_t n_listed = mode == 2 ? min(*rl.n, rl.cap) : listed_pairs(sel, order_cnt);
    if (blockIdx.x * blockDim.x >= n_listed) return; // (uniform over the block)
    stage_ends(cx.ix, ends);
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    bool in = slot < n_listed;
    const uint32_t local = in ? (mode == 2 ? rl.ids[slot] : (order ? order[slot] : slot)) : 0u;
    // (the rescue list's pairs are mode 2's: told by a flag the clustering set and nobody clears — the rescue, which runs beside mode 1,
    //  rewrites those pairs' headers and candidates meanwhile)
    if (in && mode == 1 && (pair_state(cx.state, cx.lay, cx.caps, local).hdr->flags & kAwaitRescue)) in = false;
    int nj = 0;
    uint32_t fl = 0, n_mask = 0; // n_mask: bit s = read s of the pair holds a byte that is not ACGT (a DP problem of it says so: DpJob::score on its way in)
    if (in) {
        ReadRef rd[2];
        make_reads(cx, rb, sel_pair(sel, local), rd);
        n_mask = (rd[0].codes ? 0u : 1u) | ((cx.pm.paired && !rd[1].codes) ? 2u : 0u);
        nj = stage_build(cx, local, rd, &fl);
    }
    const bool over = late.ids && (fl & kOvAny) && !(fl & kDispatched);
    if (__ballot(over)) { // (a handful of pairs per batch)
        if (over) {
            const uint32_t at = atomicAdd(late.n, 1u);
            // past the room kept for them the pair stays undispatched: k_finish lists it and it is mapped after the pass
            if (at < late.cap) { late.ids[at] = sel_pair(sel, local); late.est[at] = sel.est[local]; pair_state(cx.state, cx.lay, cx.caps, local).hdr->flags = fl | kDispatched; }
        }
    }
    // (the per-class counters are updated under a compile-time index: an array indexed by a run-time class lives in scratch memory)
    uint32_t per_class[kDpClasses] = {0, 0, 0, 0, 0, 0}, my_cells = 0, bad = 0;
    for (int k = 0; k < nj; k++) {
        const DpJob j = pair_job(cx, local, k);
        const int c = job_class(j);
        if (c < 0) bad++; else my_cells += (uint32_t)(j.rLen * j.gLen);
#pragma unroll
        for (int q = 0; q < kDpClasses; q++) per_class[q] += c == q ? 1u : 0u;
    }
    uint32_t base[kDpClasses];
#pragma unroll
    for (int c = 0; c < kDpClasses; c++) base[c] = wave_reserve(sinks.s[c].count, per_class[c]);
    for (int k = 0; k < nj; k++) {
        DpJob j = pair_job(cx, local, k);
        j.score = (int32_t)((n_mask >> j.slot) & 1u);
        const int c = job_class(j);
        if (c < 0) continue;
        uint32_t at = 0;
#pragma unroll
        for (int q = 0; q < kDpClasses; q++) if (c == q) { at = base[q]++; if (at < sinks.s[q].cap) sinks.s[q].jobs[at] = j; }
    }
    for (int o = 32; o > 0; o >>= 1) { my_cells += __shfl_down(my_cells, o, 64); bad += __shfl_down(bad, o, 64); }
    // (64 bits: the problems of one pass of BASELINE config 5 hold 5 x 10^10 cells)
    if ((threadIdx.x & 63) == 0) { if (my_cells) atomicAdd((unsigned long long *)cells, (unsigned long long)my_cells); if (bad) atomicAdd(unsupported, bad); }
}

// ---- the large tier's build: a wavefront per pair ---------------------------------------------------------------------------
// The large tier's pairs come from repeats: hundreds of candidates each, and k_build gives a pair to ONE lane, which builds them one
// after the other — the launch is as long as that lane (1.9 ms for 30 k pairs: the step waited for it).  Here a wavefront takes the
// pair and a lane a candidate, sixty-four at a time in the order stage_build goes through them:
//   * scores and mates of both reads' candidates in LDS: keep_top_scores / mask_unpaired with wave-wide maxima;
//   * a candidate's fragments go where an exclusive scan over the candidates' bounds (2 seeds + 2) puts them — holes of a fragment
//     or two instead of the serial running count: nothing reads the pool but through a candidate's (frag_off, n_frags) —, its gap
//     fragments are classified as ProcessNormalPair does, its DP problems counted by list;
//   * scans over those counts give every lane its stretch of the pair's column area (64 bytes for the summary + the columns, in
//     multiples of 8) and its places in the pass's job lists (one atomic per list and round), and the lane writes its problems there.
// Same fragments, same kinds, same DP problems as stage_build; only where they lie in the pair's pools differs, which no result sees.
// mode: as k_build's.
// lane_limit: the pairs whose bounds come to more than this are built by one lane as well (tests: 0 sends every pair that way)
__global__ void __launch_bounds__(64) k_build_wave(Ctx cx, ReadBatch rb, PairSel sel, JobSinks sinks, uint32_t *cells, uint32_t *unsupported, int mode, RescueList rl, int lane_limit)
{
    extern __shared__ int32_t bw_lds[]; // score[2][cand_cap], mate[2][cand_cap], seeds[2][cand_cap]
    __shared__ EndsLds ends;
    stage_ends(cx.ix, ends);
    const int lane = threadIdx.x;
    const int nr = cx.pm.paired ? 2 : 1, cap = cx.caps.cand_cap;
    int32_t *sc[2] = {bw_lds, bw_lds + cap}, *mt[2] = {bw_lds + 2 * cap, bw_lds + 3 * cap}, *cn[2] = {bw_lds + 4 * cap, bw_lds + 5 * cap};
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_s_barrier(); };
    auto wave_max = [](int v) { for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64)); return v; };
    auto wave_sum = [](uint32_t v) { for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64); return v; };
    auto excl_scan = [&](uint32_t v) { uint32_t in = v; for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)in, o, 64); if (lane >= o) in += t; } return in - v; };
    const uint32_t n_listed = mode == 2 ? min(*rl.n, rl.cap) : sel.n;
    uint32_t my_cells = 0, bad = 0;
    for (uint32_t slot = blockIdx.x; slot < n_listed; slot += gridDim.x) {
        const uint32_t local = mode == 2 ? rl.ids[slot] : slot;
        PairState st = pair_state(cx.state, cx.lay, cx.caps, local);
        PairHdr h = *st.hdr; // (every lane the same words: one fetch)
        if (mode == 1 && (h.flags & kAwaitRescue)) continue;
        auto put_hdr = [&](uint32_t flags, int n_frags, int n_ops, int n_jobs) {
            if (lane == 0) { st.hdr->flags = flags; st.hdr->n_frags = n_frags; st.hdr->n_ops = n_ops; st.hdr->n_jobs = (int16_t)n_jobs; }
        };
        if (h.flags & kOvAny) { put_hdr(h.flags, 0, 0, 0); continue; }
        const int n_c[2] = {h.n_cands[0], nr == 2 ? h.n_cands[1] : 0};
        ReadRef rd[2];
        make_reads(cx, rb, sel_pair(sel, local), rd);
        // ---- scores and mates into LDS; the masks of ReadMapping.cpp:469-470 (keep_top_scores / mask_unpaired)
        wave_sync();
        for (int s = 0; s < nr; s++) for (int i = lane; i < n_c[s]; i += 64) { const Cand c = st.cands[s][i]; sc[s][i] = c.score; mt[s][i] = c.mate; cn[s][i] = c.count; }
        wave_sync();
        auto keep_top = [&](int s) {
            if (n_c[s] <= 1) return;
            int best = 0;
            for (int i = lane; i < n_c[s]; i += 64) best = max(best, sc[s][i]);
            best = wave_max(best);
            for (int i = lane; i < n_c[s]; i += 64) if (sc[s][i] < best) sc[s][i] = 0;
        };
        if (cx.pm.paired && h.n_paired != 0) {
            int top = 0;
            for (int i = lane; i < n_c[0]; i += 64) if (mt[0][i] != -1) top = max(top, sc[0][i] + sc[1][mt[0][i]]);
            top = wave_max(top);
            for (int i = lane; i < n_c[0]; i += 64) if (mt[0][i] == -1 || sc[0][i] + sc[1][mt[0][i]] < top) sc[0][i] = 0; // (read 2's scores are still what they were)
            wave_sync();
            for (int j = lane; j < n_c[1]; j += 64) if (mt[1][j] == -1 || sc[1][j] + sc[0][mt[1][j]] < top) sc[1][j] = 0;  // (read 1's as the loop above left them)
        } else { keep_top(0); if (cx.pm.paired) keep_top(1); }
        wave_sync();
        // ---- the fragments are placed by their bounds (2 seeds + 2 per live candidate), stage_build places them densely: when the bounds of
        //      all live candidates fit the pool both ways fit; when they do not, the pair is built stage_build's way, by one lane — the same
        //      fragments wherever they lie, and the same answer to "does this pair fit the tier" (a pair that does not is fatal for the batch)
        {
            uint32_t tb = 0;
            for (int s = 0; s < nr; s++) for (int i = lane; i < n_c[s]; i += 64) if (sc[s][i] != 0) tb += 2u * (uint32_t)cn[s][i] + 2u;
            tb = wave_sum(tb);
            if (tb > (uint32_t)min(cx.caps.frag_cap, lane_limit)) {
                if (lane == 0) {
                    const int nj = stage_build(cx, local, rd);
                    for (int k = 0; k < nj; k++) {
                        DpJob j = pair_job(cx, local, k);
                        j.score = rd[j.slot].codes ? 0 : 1;
                        const int q = job_class(j);
                        if (q < 0) { bad++; continue; }
                        my_cells += (uint32_t)(j.rLen * j.gLen);
#pragma unroll
                        for (int c = 0; c < kDpClasses; c++) if (q == c) { const uint32_t at = atomicAdd(sinks.s[c].count, 1u); if (at < sinks.s[c].cap) sinks.s[c].jobs[at] = j; }
                    }
                }
                continue;
            }
        }
        // ---- the candidates, a lane each, in stage_build's order
        uint32_t flags = h.flags, n_frags = 0, n_ops = 0, n_jobs = 0;
        const int total = n_c[0] + n_c[1];
        for (int base = 0; base < total && !(flags & kOvAny); base += 64) {
            const int t = base + lane;
            const bool mine = t < total;
            const int s = (mine && t >= n_c[0]) ? 1 : 0, ci = mine ? t - (s ? n_c[0] : 0) : 0;
            Cand c; c.count = 0; c.first = 0; c.in_pool = 0; c.pool_off = 0; c.score = 0;
            bool live = false;
            if (mine) { c = st.cands[s][ci]; live = sc[s][ci] != 0; }
            const uint32_t bound = live ? 2u * (uint32_t)c.count + 2u : 0u;
            const uint32_t off = n_frags + excl_scan(bound);
            const uint32_t round_frags = wave_sum(bound);
            if (n_frags + round_frags > (uint32_t)cx.caps.frag_cap) { flags |= kOvFrags; break; } // (uniform)
            Frag *f = st.frags + off;
            int nf = 0;
            uint32_t my_ops = 0, my_cls[kDpClasses] = {0, 0, 0, 0, 0, 0};
            if (mine) {
                if (live) {
                    nf = build_frags(cx.ix, rd[s].rlen, c.in_pool ? cx.seed_pool + c.pool_off : st.hits[s] + c.first, c.count, f);
                    // ProcessNormalPair (:155-191): classify each gap fragment; a DP problem's place in the column area comes later
                    for (int i = 0; i < nf; i++) {
                        Frag x = f[i];
                        if (x.kind == kSimple) continue;
                        if (x.rLen > 0 && x.gLen > 0) {
                            bool dp = x.rLen != x.gLen;
                            int mm = -1;
                            if (!dp) { mm = frag_mismatches(cx.ix, x, rd[s]); dp = mm > 1 && mm >= (int)(x.rLen * 0.2); }
                            if (dp) {
                                x.kind = kDp; x.ops_off = 0; x.ops_len = 0; x.meta = 0;
                                my_ops += (uint32_t)kDpSum + (((uint32_t)(x.rLen + x.gLen) + 7u) & ~7u);
                                DpJob j; j.rLen = x.rLen; j.gLen = x.gLen;
                                const int q = job_class(j);
                                if (q < 0) bad++; else my_cells += (uint32_t)(x.rLen * x.gLen);
#pragma unroll
                                for (int k = 0; k < kDpClasses; k++) my_cls[k] += q == k ? 1u : 0u;
                            } else { x.kind = kPlain; x.ops_len = x.rLen; x.meta = (uint32_t)(mm + 1); }
                        } else if (x.rLen > 0) { x.kind = kIns; x.ops_len = x.rLen; }
                        else { x.kind = kDel; x.ops_len = x.gLen; }
                        f[i] = x;
                    }
                }
                // (what changed of the candidate: its place in the fragment pool, and its score when a mask or the validity check dropped it)
                Cand *g = st.cands[s] + ci;
                g->frag_off = (int16_t)off; g->n_frags = (int16_t)(nf < 0 ? 0 : nf);
                const int new_score = (live && nf >= 0) ? c.score : 0;
                if (new_score != c.score) g->score = new_score;
            }
            // ---- places: the column area of the pair, the job lists of the pass
            uint32_t my_jobs = 0;
#pragma unroll
            for (int k = 0; k < kDpClasses; k++) my_jobs += my_cls[k];
            uint32_t cur = n_ops + excl_scan(my_ops);
            const uint32_t round_ops = wave_sum(my_ops), round_jobs = wave_sum(my_jobs);
            if (n_ops + round_ops > (uint32_t)cx.caps.ops_cap) { flags |= kOvOps; break; }
            if (n_jobs + round_jobs > (uint32_t)cx.caps.job_cap) { flags |= kOvJobs; break; }
            uint32_t at_cls[kDpClasses];
#pragma unroll
            for (int k = 0; k < kDpClasses; k++) at_cls[k] = wave_reserve(sinks.s[k].count, my_cls[k]);
            if (my_jobs) {
                for (int i = 0; i < nf; i++) {
                    Frag x = f[i];
                    if (x.kind != kDp) continue;
                    x.ops_off = (int64_t)(cur + (uint32_t)kDpSum);
                    cur += (uint32_t)kDpSum + (((uint32_t)(x.rLen + x.gLen) + 7u) & ~7u);
                    f[i] = x;
                    DpJob j;
                    j.pair = local; j.slot = (uint16_t)s; j.rev = x.gPos >= cx.ix.G ? 1 : 0;
                    j.rPos = x.rPos; j.rLen = x.rLen; j.gPos = x.gPos; j.gLen = x.gLen;
                    j.ops_off = (int32_t)x.ops_off; j.frag = (int32_t)(off + (uint32_t)i); j.score = rd[s].codes ? 0 : 1;
                    const int q = job_class(j);
#pragma unroll
                    for (int k = 0; k < kDpClasses; k++) if (q == k) { const uint32_t at = at_cls[k]++; if (at < sinks.s[k].cap) sinks.s[k].jobs[at] = j; }
                }
            }
            n_frags += round_frags; n_ops += round_ops; n_jobs += round_jobs;
        }
        put_hdr(flags, (int)n_frags, (int)n_ops, (int)n_jobs);
    }
    my_cells = wave_sum(my_cells); bad = wave_sum(bad);
    if (lane == 0) { if (my_cells) atomicAdd((unsigned long long *)cells, (unsigned long long)my_cells); if (bad) atomicAdd(unsupported, bad); }
}

// LDS per problem is sized per class: the small classes are latency-bound (a chain of dependent
// fetches per problem), so what counts is how many problems a CU holds at once; the rare problem
// that does not fit its class's LDS keeps its sequences / traceback in the workgroup's HBM scratch.
template <int K> struct DpLds { static constexpr int seq = kDpLdsSeq, dir = kDpLdsDir; };
template <> struct DpLds<1> { static constexpr int seq = 512, dir = 4096; }; // targets <= 64: e.g. 48 x 48 fits
template <> struct DpLds<4> { static constexpr int seq = 1024, dir = 3072; }; // targets 65..256: the traceback of most does not fit 12 KB either — more problems per CU instead

// one problem on the W lanes of a group (W = 64: the wave; 32: a half wave): stage the two strings,
// sweep, trace back, hand the column string to the fragment
template <int K, int W>
static __device__ __forceinline__ void dp_run_job(const Ctx &cx, const JobSink &sink, uint32_t jb, const DpJob &job, const ReadBatch &rb,
                                                  const PairSel &sel, const DpBuf &b)
{
    const int nr = cx.pm.paired ? 2 : 1;
    const int lane = threadIdx.x & (W - 1);
    const uint32_t read = sel_pair(sel, job.pair) * nr + job.slot;
    ReadRef rd;
    rd.ascii = rb.bases + rb.off[read]; rd.rlen = (int)(rb.off[read + 1] - rb.off[read]); rd.flipped = (cx.pm.paired && job.slot == 1) ? 1 : 0;
    // q = read fragment, t = genome fragment; both reversed on the reverse strand (the
    // reference also complements both, which no comparison can see)
    for (int i = lane; i < job.rLen; i += W) b.q[i] = (uint8_t)read_code(rd, job.rev ? job.rPos + job.rLen - 1 - i : job.rPos + i);
    for (int i = lane; i < job.gLen; i += W) b.t[i] = (uint8_t)ref_code(cx.ix, job.rev ? job.gPos + job.gLen - 1 - i : job.gPos + i);
    dp_sync<W>();
    PairState st = pair_state(cx.state, cx.lay, cx.caps, job.pair);
    int score = 0;
    DpSummary *sum = cx.dp_summary ? (DpSummary *)(st.ops + job.ops_off - kDpSum) : nullptr; // (stage_build left room for it)
    const int w = dp_core<K, W>(cx.pm.use_nw != 0, job.rLen, job.gLen, b, st.ops + job.ops_off, &score, sum, (uint32_t)job.ops_off);
    if (lane == 0) {
        Frag f = st.frags[job.frag]; // one fetch, one store (the fields share two words)
        f.ops_off = job.ops_off + w;
        f.ops_len = job.rLen + job.gLen - w;
        f.meta = sum ? (uint32_t)((job.ops_off - kDpSum) >> 3) + 1u : 0u;
        st.frags[job.frag] = f;
        sink.jobs[jb].score = score;
    }
    dp_sync<W>();
}

template <int K>
__global__ void __launch_bounds__(64) k_dp_sel(Ctx cx, JobSink sink, ReadBatch rb, PairSel sel, uint8_t *scratch,
                                               uint64_t scratch_stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[DpLds<K>::seq + DpLds<K>::dir];
    uint8_t *spill = scratch + (uint64_t)blockIdx.x * scratch_stride;
    const uint32_t n = min(*sink.count, sink.cap);
    for (uint32_t jb = blockIdx.x; jb < n; jb += gridDim.x) {
        const DpJob job = sink.jobs[jb];
        dp_run_job<K, 64>(cx, sink, jb, job, rb, sel, dp_buffers(job.rLen, job.gLen, lds, spill, DpLds<K>::seq, DpLds<K>::dir));
    }
}

// The one-wavefront classes, a group of problems at a time.  k_dp_sel sweeps a problem and then lets lane 0 walk its traceback
// while 63 lanes look on — as many vector instructions as the sweep itself.  Here a wavefront sweeps up to 64 problems one after
// the other, every sweep leaving its traceback bytes (and the two strings) in the wavefront's stretch of an HBM scratch that
// stays in L2, and then walks the 64 tracebacks at once, one per lane.  Same bytes, same walks, same column strings.
constexpr int kDpGroup = 64;
struct DpGroupSlot { uint32_t off; int32_t score; }; // where a problem's strings and traceback bytes lie in the wave's scratch; its sweep's score

template <int K>
__global__ void __launch_bounds__(64) k_dp_group(Ctx cx, JobSink sink, ReadBatch rb, PairSel sel, uint8_t *scratch, uint64_t scratch_stride, uint32_t max_n)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[DpLds<K>::seq];
    __shared__ DpGroupSlot slot[kDpGroup];
    uint8_t *mine = scratch + (uint64_t)blockIdx.x * scratch_stride;
    const uint32_t n = min(*sink.count, sink.cap);
    if (n >= max_n) return; // (a long list: k_dp_lane's)
    const int nr = cx.pm.paired ? 2 : 1;
    const int lane = threadIdx.x;
    const bool nw = cx.pm.use_nw != 0;
    // (few problems: small groups, so that every wavefront of the launch gets some; many: whole groups of 64)
    uint32_t group = (n + 2 * gridDim.x - 1) / (2 * gridDim.x);
    group = group < 8 ? 8 : (group > (uint32_t)kDpGroup ? (uint32_t)kDpGroup : group);
    for (uint32_t slice = blockIdx.x * group; slice < n; slice += gridDim.x * group) {
        const uint32_t slice_end = min(slice + group, n);
        for (uint32_t jb0 = slice; jb0 < slice_end;) {
            // ---- the sweeps, one problem after the other, all lanes on each; the group ends when the wave's stretch of scratch is full ----
            uint32_t used = 0;
            int g_n = 0;
            for (; jb0 + g_n < slice_end; g_n++) {
                const DpJob job = sink.jobs[jb0 + g_n];
                const uint32_t need = (uint32_t)((job.rLen + job.gLen + 15) & ~15) + (uint32_t)(job.rLen + job.gLen - 1) * (uint32_t)job.gLen;
                if (g_n > 0 && used + need > scratch_stride) break; // (a stretch holds the largest problem of its class)
                const uint32_t read = sel_pair(sel, job.pair) * nr + job.slot;
                ReadRef rd;
                rd.ascii = rb.bases + rb.off[read]; rd.rlen = (int)(rb.off[read + 1] - rb.off[read]); rd.flipped = (cx.pm.paired && job.slot == 1) ? 1 : 0;
                uint8_t *gq = mine + used, *gt = gq + job.rLen, *gdir = gq + ((job.rLen + job.gLen + 15) & ~15);
                DpBuf b;
                const bool in_lds = job.rLen <= DpLds<K>::seq / 2 && job.gLen <= DpLds<K>::seq / 2;
                b.q = in_lds ? lds : gq; b.t = in_lds ? lds + DpLds<K>::seq / 2 : gt; b.dir = gdir;
                for (int i = lane; i < job.rLen; i += 64) { const uint8_t c = (uint8_t)read_code(rd, job.rev ? job.rPos + job.rLen - 1 - i : job.rPos + i); b.q[i] = c; if (in_lds) gq[i] = c; }
                for (int i = lane; i < job.gLen; i += 64) { const uint8_t c = (uint8_t)ref_code(cx.ix, job.rev ? job.gPos + job.gLen - 1 - i : job.gPos + i); b.t[i] = c; if (in_lds) gt[i] = c; }
                __syncthreads();
                int score = 0;
                dp_sweep<K, 64>(nw, job.rLen, job.gLen, b, &score);
                if (lane == 0) { slot[g_n].off = used; slot[g_n].score = score; }
                used += need;
                __syncthreads();
            }
            __threadfence_block();
            __syncthreads();
            // ---- the walks, one problem per lane ----
            if (lane < g_n) {
                const uint32_t jb = jb0 + (uint32_t)lane;
                const DpJob job = sink.jobs[jb];
                const uint8_t *gq = mine + slot[lane].off, *gt = gq + job.rLen, *gdir = gq + ((job.rLen + job.gLen + 15) & ~15);
                PairState st = pair_state(cx.state, cx.lay, cx.caps, job.pair);
                DpSummary *sum = cx.dp_summary ? (DpSummary *)(st.ops + job.ops_off - kDpSum) : nullptr;
                const int w = dp_trace(nw, job.rLen, job.gLen, gq, gt, gdir, st.ops + job.ops_off, sum, (uint32_t)job.ops_off);
                Frag f = st.frags[job.frag];
                f.ops_off = job.ops_off + w;
                f.ops_len = job.rLen + job.gLen - w;
                f.meta = sum ? (uint32_t)((job.ops_off - kDpSum) >> 3) + 1u : 0u;
                st.frags[job.frag] = f;
                sink.jobs[jb].score = slot[lane].score;
            }
            __syncthreads();
            jb0 += (uint32_t)g_n;
        }
    }
}

constexpr int kDpHalfT = 32, kDpHalfQ = 64, kDpHalfLds = kDpHalfQ + kDpHalfT + (kDpHalfQ + kDpHalfT - 1) * kDpHalfT + 32;
// One problem per LANE (mcx_dp_lane.h): a wavefront takes 64 problems of its list at a time; the group's words (queries, strip
// edges, packed traceback flags) lie lane-interleaved in the wavefront's stretch of scratch, laid out for the group's longest
// query and widest target, so that every store and load of the sweep is one line per 16 lanes.  order: the list's problems by
// size (null: as listed), so that the 64 of a group finish together.
template <int K, bool NW>
__global__ void __launch_bounds__(64) k_dp_lane(Ctx cx, JobSink sink, const uint32_t *order, ReadBatch rb, PairSel sel, uint32_t *scratch, uint64_t stride_words, uint32_t *unsupported,
                                                uint32_t min_n)
{
    const uint32_t n = min(*sink.count, sink.cap);
    if (n < min_n) return; // (a short list: k_dp_group's)
    const int lane = threadIdx.x;
    const int nr = cx.pm.paired ? 2 : 1;
    LaneMem mem; mem.base = scratch + (uint64_t)blockIdx.x * stride_words; mem.stride = 64; mem.lane = (uint32_t)lane;
    for (uint32_t g0 = blockIdx.x * 64u; g0 < n; g0 += gridDim.x * 64u) {
        const uint32_t jb = g0 + (uint32_t)lane;
        const bool have = jb < n;
        const uint32_t at = have ? (order ? order[jb] : jb) : 0u;
        DpJob job;
        if (have) job = sink.jobs[at]; else { job.rLen = 0; job.gLen = 0; }
        int rows = job.rLen, strips = (job.gLen + K - 1) / K;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { rows = max(rows, __shfl_xor(rows, o, 64)); strips = max(strips, __shfl_xor(strips, o, 64)); }
        const LaneLayout l = lane_layout<K, NW>(rows, strips);
        if ((uint64_t)l.words * 64u > stride_words) { if (lane == 0) atomicAdd(unsupported, 1u); continue; } // (cannot happen: the lists' size limits are the strides')
        if (!have) continue;
        const uint32_t read = sel_pair(sel, job.pair) * nr + job.slot;
        ReadRef rd;
        rd.ascii = rb.bases + rb.off[read]; rd.rlen = (int)(rb.off[read + 1] - rb.off[read]); rd.flipped = (cx.pm.paired && job.slot == 1) ? 1 : 0;
        rd.codes = (cx.packed && !(cx.read_ext[read] >> 31)) ? cx.packed + (uint64_t)read * cx.wpad : nullptr;
        sink.jobs[at].score = lane_dp_job<K, NW>(cx, mem, l, job, rd);
    }
}

// TWO problems per lane (mcx_dp_lane2.h): a wavefront takes 128 problems of its list at a time, lane l the neighbours 2l and 2l + 1 of the (shape-sorted)
// list; every value of both recurrences in the sixteen bits it needs, problem A in the low half of a register and problem B in the high one, so that one
// v_pk_*_i16 instruction advances both — the arithmetic width of the reference's own vectors (ksw2_alignment.cpp:70-248: sixteen int8 lanes).  Same words
// per problem in the wavefront's stretch of scratch, same column strings and summaries as k_dp_lane (MCX_DP_X1=1 runs that one: the A/B of the parity tests).
template <int K, bool NW>
__global__ void __launch_bounds__(64) k_dp_lane2(Ctx cx, JobSink sink, const uint32_t *order, ReadBatch rb, PairSel sel, uint32_t *scratch, uint64_t stride_words, uint32_t *unsupported,
                                                 uint32_t min_n)
{
    const uint32_t n = min(*sink.count, sink.cap);
    if (n < min_n) return; // (a short list: k_dp_group's)
    const int lane = threadIdx.x;
    const int nr = cx.pm.paired ? 2 : 1;
    LaneMem mem; mem.base = scratch + (uint64_t)blockIdx.x * stride_words; mem.stride = 64; mem.lane = (uint32_t)lane;
    for (uint32_t g0 = blockIdx.x * 128u; g0 < n; g0 += gridDim.x * 128u) {
        const uint32_t ja = g0 + 2u * (uint32_t)lane, jb = ja + 1u;
        const bool have_a = ja < n, have_b = jb < n;
        const uint32_t at_a = have_a ? (order ? order[ja] : ja) : 0u, at_b = have_b ? (order ? order[jb] : jb) : at_a;
        DpJob job_a, job_b;
        if (have_a) job_a = sink.jobs[at_a]; else { job_a.rLen = 0; job_a.gLen = 0; }
        if (have_b) job_b = sink.jobs[at_b]; else job_b = job_a;
        int rows = max(job_a.rLen, job_b.rLen), strips = (max(job_a.gLen, job_b.gLen) + K - 1) / K;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { rows = max(rows, __shfl_xor(rows, o, 64)); strips = max(strips, __shfl_xor(strips, o, 64)); }
        const LaneLayout2 l = lane_layout2<K, NW>(rows, strips);
        if ((uint64_t)l.words * 64u > stride_words) { if (lane == 0) atomicAdd(unsupported, 1u); continue; } // (cannot happen: the lists' size limits are the strides')
        if (!have_a) continue;
        ReadRef rd[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const DpJob &job = h ? job_b : job_a;
            const uint32_t read = sel_pair(sel, job.pair) * nr + job.slot;
            rd[h].flipped = (cx.pm.paired && job.slot == 1) ? 1 : 0;
            // (the job says whether its read holds an N — k_build knew —: a read without one is its 2-bit words and nothing else, no look at its offsets or its flags)
            if (cx.packed && job.score == 0) { rd[h].codes = cx.packed + (uint64_t)read * cx.wpad; rd[h].ascii = nullptr; rd[h].rlen = 0; }
            else { rd[h].codes = nullptr; rd[h].ascii = rb.bases + rb.off[read]; rd[h].rlen = (int)(rb.off[read + 1] - rb.off[read]); }
        }
        int sc[2];
        lane_dp_job2<K, NW>(cx, mem, l, job_a, rd[0], have_b, job_b, rd[1], sc);
        sink.jobs[at_a].score = sc[0];
        if (have_b) sink.jobs[at_b].score = sc[1];
    }
}

// the three short lists (job_class): tiny <= 8 x 8 in strips of 8; small: targets <= 16, queries <= 32; half: targets <= 32, queries <= 64
static uint64_t lane_short_words(int which) // ksw2's flags take more words than nw's: sized for them; a lane of k_dp_lane2 keeps two problems
{
    const uint64_t one = which == 0 ? 64ull * lane_layout<8, false>(kDpTiny, 1).words : which == 1 ? 64ull * lane_layout<16, false>(kDpSmallQ, 1).words : 64ull * lane_layout<16, false>(kDpHalfQ, 2).words;
    const uint64_t two = which == 0 ? 64ull * lane_layout2<8, false>(kDpTiny, 1).words : which == 1 ? 64ull * lane_layout2<16, false>(kDpSmallQ, 1).words : 64ull * lane_layout2<16, false>(kDpHalfQ, 2).words;
    return std::max(one, two);
}

// ---- the problems of a long list by shape ------------------------------------------------------------------------------------
// A wavefront of k_dp_lane runs as long as the longest query times the most strips among its problems.  The two long lists
// (targets of 17-64 and of 65-256 bases, queries of any length) are therefore dealt to the wavefronts by shape: 1024 buckets of
// (strips, query length in 64 classes), largest first; within a bucket the problems differ by less than 4 rows for reads of up
// to 256 bases (8 / 16 rows for longer ones).  With the 16 row classes this began with, a wavefront's rows were the class's
// largest — 7.5 rows above its problems' mean, a sixth of the cells of config 5's 45-row problems computed for nothing.  Three
// small passes — count per bucket, start of every bucket, place — over the list's 40-byte records; the order among equals is
// whatever the atomics give (no result depends on it).
constexpr int kDpBuckets = 1024, kDpRowClasses = 64, kDpSortTile = 8;
static __device__ __forceinline__ int dp_bucket(const DpJob &j, int row_shift)
{
    const int strips = (j.gLen + 15) >> 4, rows = min(kDpRowClasses - 1, j.rLen >> row_shift);
    return (min(16, max(strips, 1)) - 1) * kDpRowClasses + rows; // (0..1023; the largest shapes get the largest numbers)
}

__global__ void __launch_bounds__(256) k_dp_sort_count(JobSink sink, int row_shift, uint32_t *counts, uint32_t min_n)
{
    __shared__ uint32_t h[kDpBuckets];
    const uint32_t n = min(*sink.count, sink.cap);
    if (n < min_n) return;
    for (int b = threadIdx.x; b < kDpBuckets; b += 256) h[b] = 0u;
    __syncthreads();
    for (uint32_t base = blockIdx.x * (256u * kDpSortTile); base < n; base += gridDim.x * (256u * kDpSortTile))
        for (int t = 0; t < kDpSortTile; t++) {
            const uint32_t i = base + t * 256u + threadIdx.x;
            if (i < n) atomicAdd(&h[dp_bucket(sink.jobs[i], row_shift)], 1u);
        }
    __syncthreads();
    for (int b = threadIdx.x; b < kDpBuckets; b += 256) if (h[b]) atomicAdd(&counts[b], h[b]);
}

// counts[0..1024) -> cursor[b] = where bucket b begins when the buckets are laid out from the largest shape down
__global__ void __launch_bounds__(256) k_dp_sort_scan(const uint32_t *counts, uint32_t *cursor)
{
    __shared__ uint32_t c[kDpBuckets], part[256];
    constexpr int per = kDpBuckets / 256;
    // thread t owns the buckets kDpBuckets-1 - (per t .. per t + per-1): the largest shapes first
    uint32_t mine[per], sum = 0;
    for (int k = 0; k < per; k++) { mine[k] = counts[kDpBuckets - 1 - (per * (int)threadIdx.x + k)]; sum += mine[k]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t at = 0; for (int k = 0; k < 256; k++) { const uint32_t m = part[k]; part[k] = at; at += m; } }
    __syncthreads();
    uint32_t at = part[threadIdx.x];
    for (int k = 0; k < per; k++) { c[per * threadIdx.x + k] = at; at += mine[k]; }
    for (int k = 0; k < per; k++) cursor[kDpBuckets - 1 - (per * (int)threadIdx.x + k)] = c[per * threadIdx.x + k];
}

__global__ void __launch_bounds__(256) k_dp_sort_place(JobSink sink, int row_shift, uint32_t *cursor, uint32_t *order, uint32_t min_n)
{
    __shared__ uint32_t h[kDpBuckets], at[kDpBuckets];
    const uint32_t n = min(*sink.count, sink.cap);
    if (n < min_n) return;
    for (uint32_t base = blockIdx.x * (256u * kDpSortTile); base < n; base += gridDim.x * (256u * kDpSortTile)) {
        for (int b = threadIdx.x; b < kDpBuckets; b += 256) h[b] = 0u;
        __syncthreads();
        int b[kDpSortTile];
        uint32_t rank[kDpSortTile];
#pragma unroll
        for (int t = 0; t < kDpSortTile; t++) {
            const uint32_t i = base + t * 256u + threadIdx.x;
            b[t] = i < n ? dp_bucket(sink.jobs[i], row_shift) : -1;
            rank[t] = b[t] >= 0 ? atomicAdd(&h[b[t]], 1u) : 0u;
        }
        __syncthreads();
        for (int q = threadIdx.x; q < kDpBuckets; q += 256) at[q] = h[q] ? atomicAdd(&cursor[q], h[q]) : 0u;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < kDpSortTile; t++) if (b[t] >= 0) order[at[b[t]] + rank[t]] = base + t * 256u + threadIdx.x;
        __syncthreads();
    }
}

// words a wavefront's stretch of scratch must hold for a list whose problems have at most `rows` query bases and `strips` strips
template <int K>
static uint64_t lane_stride_words(bool nw, int rows, int strips, bool x2)
{
    if (x2) return 64ull * (nw ? lane_layout2<K, true>(rows, strips).words : lane_layout2<K, false>(rows, strips).words);
    return 64ull * (nw ? lane_layout<K, true>(rows, strips).words : lane_layout<K, false>(rows, strips).words);
}

template <int K>
static void launch_dp_lane(bool nw, bool x2, unsigned blocks, hipStream_t s, const Ctx &cx, const JobSink &sink, const uint32_t *order, const ReadBatch &rb, const PairSel &sel,
                           uint32_t *scratch, uint64_t stride_words, uint32_t *unsupported, uint32_t min_n)
{
    if (x2) {
        if (nw) k_dp_lane2<K, true><<<blocks, 64, 0, s>>>(cx, sink, order, rb, sel, scratch, stride_words, unsupported, min_n);
        else k_dp_lane2<K, false><<<blocks, 64, 0, s>>>(cx, sink, order, rb, sel, scratch, stride_words, unsupported, min_n);
        return;
    }
    if (nw) k_dp_lane<K, true><<<blocks, 64, 0, s>>>(cx, sink, order, rb, sel, scratch, stride_words, unsupported, min_n);
    else k_dp_lane<K, false><<<blocks, 64, 0, s>>>(cx, sink, order, rb, sel, scratch, stride_words, unsupported, min_n);
}

// targets <= 32 (queries <= 64) of the one-column-per-lane class: two problems per wave, 32 lanes each — the
// class is bound by vector instructions issued, and most of its targets are that short

__global__ void __launch_bounds__(256) k_dp_half(Ctx cx, JobSink sink, ReadBatch rb, PairSel sel)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[8 * kDpHalfLds];
    const int group = threadIdx.x >> 5;
    uint8_t *mine = lds + group * kDpHalfLds;
    DpBuf b; b.q = mine; b.t = mine + kDpHalfQ; b.dir = mine + kDpHalfQ + kDpHalfT;
    const uint32_t n = min(*sink.count, sink.cap);
    for (uint32_t jb = blockIdx.x * 8 + group; jb < n; jb += gridDim.x * 8)
        dp_run_job<1, 32>(cx, sink, jb, sink.jobs[jb], rb, sel, b);
}

// one tiny problem per lane (mcx_dp.h kDpTiny)
__global__ void __launch_bounds__(256) k_dp_tiny(Ctx cx, JobSink sink, ReadBatch rb, PairSel sel)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[256 * kDpTinyLds];
    uint8_t *mine = lds + threadIdx.x * kDpTinyLds;
    DpBuf b; b.q = mine; b.t = mine + kDpTiny; b.dir = mine + 2 * kDpTiny;
    const uint32_t n = min(*sink.count, sink.cap);
    for (uint32_t jb = blockIdx.x * blockDim.x + threadIdx.x; jb < n; jb += gridDim.x * blockDim.x)
        dp_run_job<kDpTiny, 1>(cx, sink, jb, sink.jobs[jb], rb, sel, b);
}

// four small problems per wave, sixteen per block; each 16-lane group owns 800 bytes of LDS
__global__ void __launch_bounds__(256) k_dp_small(Ctx cx, JobSink sink, ReadBatch rb, PairSel sel)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[16 * kDpSmallLds];
    const int nr = cx.pm.paired ? 2 : 1;
    const int group = threadIdx.x >> 4, lane = threadIdx.x & 15;
    uint8_t *mine = lds + group * kDpSmallLds;
    const uint32_t n = min(*sink.count, sink.cap);
    for (uint32_t jb = blockIdx.x * 16 + group; jb < n; jb += gridDim.x * 16) {
        const DpJob job = sink.jobs[jb];
        const uint32_t read = sel_pair(sel, job.pair) * nr + job.slot;
        ReadRef rd;
        rd.ascii = rb.bases + rb.off[read]; rd.rlen = (int)(rb.off[read + 1] - rb.off[read]); rd.flipped = (cx.pm.paired && job.slot == 1) ? 1 : 0;
        DpBuf b; b.q = mine; b.t = mine + kDpSmallQ; b.dir = mine + 64;
        for (int i = lane; i < job.rLen; i += 16) b.q[i] = (uint8_t)read_code(rd, job.rev ? job.rPos + job.rLen - 1 - i : job.rPos + i);
        if (lane < job.gLen) b.t[lane] = (uint8_t)ref_code(cx.ix, job.rev ? job.gPos + job.gLen - 1 - lane : job.gPos + lane);
        dp_sync<16>();
        PairState st = pair_state(cx.state, cx.lay, cx.caps, job.pair);
        int score = 0;
        DpSummary *sum = cx.dp_summary ? (DpSummary *)(st.ops + job.ops_off - kDpSum) : nullptr;
        const int w = dp_core<1, 16>(cx.pm.use_nw != 0, job.rLen, job.gLen, b, st.ops + job.ops_off, &score, sum, (uint32_t)job.ops_off);
        if (lane == 0) {
            Frag f = st.frags[job.frag]; // one fetch, one store (the fields share two words)
            f.ops_off = job.ops_off + w;
            f.ops_len = job.rLen + job.gLen - w;
            f.meta = sum ? (uint32_t)((job.ops_off - kDpSum) >> 3) + 1u : 0u;
            st.frags[job.frag] = f;
            sink.jobs[jb].score = score;
        }
        dp_sync<16>();
    }
}

#ifndef MCX_FINISH_WAVES
#define MCX_FINISH_WAVES 4 // (109 registers, no spills; five waves at 96 spilled 16)
#endif
__global__ void __launch_bounds__(256, MCX_FINISH_WAVES) k_finish(Ctx cx, ReadBatch rb, PairSel sel, AlnRec *recs, PairOut *pout, uint32_t *ov_ids, uint32_t *n_ov,
                                                uint32_t ov_cap, uint32_t *pool_over, const uint32_t *order, const uint32_t *order_cnt)
{
    __shared__ EndsLds ends;
    __shared__ uint32_t cig_stage[256 * 2 * kCigStage]; // the first operations of every read, word-major: word k of thread t at [k * 256 + t] (neighbouring lanes, neighbouring banks)
    if (blockIdx.x * blockDim.x >= listed_pairs(sel, order_cnt)) return; // (uniform over the block)
    stage_ends(cx.ix, ends);
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t *stage = cig_stage + threadIdx.x;
    const bool active = slot < listed_pairs(sel, order_cnt); // (no early exit within a wave: it reserves its CIGAR words together)
    const uint32_t local = active ? (order ? order[slot] : slot) : 0u;
    const int nr = cx.pm.paired ? 2 : 1;
    uint32_t pair = 0;
    ReadRef rd[2];
    PairState st;
    PairHdr h; // the final header stays in registers: nothing reads the pair state after this kernel
    h.flags = 0;
    int n_cig[2] = {0, 0};
    uint8_t *detail2 = nullptr;
    if (active) {
        pair = sel_pair(sel, local);
        make_reads(cx, rb, pair, rd);
        st = pair_state(cx.state, cx.lay, cx.caps, local);
        h = *st.hdr;
        st.hdr = &h;
        detail2 = cx.detail ? cx.detail + (int64_t)pair * nr * cx.dlay.stride : nullptr; // records are indexed by batch read
        finish_scores(cx, st, rd, (DetailHdr *)detail2, n_cig, nullptr, stage, 256);
    }
    const uint32_t want = (uint32_t)(n_cig[0] + n_cig[1]);
    const uint32_t at = wave_reserve(cx.cig_pool_n, want);
    if (!active || (h.flags & kDispatched)) return; // (a dispatched pair is the large tier's: its records and summary come from there)
    const bool fits = at + want <= cx.cig_pool_cap;
    if (!fits) atomicOr(pool_over, 1u);
    const uint32_t off[2] = {at, at + (uint32_t)n_cig[0]};
    finish_records(cx, st, rd, recs + (int64_t)pair * nr, fits ? cx.cig_pool : nullptr, off, n_cig, detail2, stage, 256);
    PairOut o;
    o.flags = h.flags; o.est = h.est; o.est_lo = h.est_lo; o.est_hi = h.est_hi;
    o.pair_dist = h.pair_dist; o.pair_ok = (int16_t)h.pair_ok; o.mapped = (int16_t)h.mapped;
    pout[pair] = o;
    if (h.flags & kOvAny) {
        const uint32_t at2 = atomicAdd(n_ov, 1u);
        if (at2 < ov_cap) ov_ids[at2] = pair;
    }
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
// The MCX_* switches of DESIGN §3 (none changes a result), read ONCE when a context is made: nothing in the launch path asks the
// environment.  What was measured slower and served no test is gone (the heavy pairs clustered first, one DP stream per list, the
// ungrouped wavefront DP, the narrow comparison windows of the seeding walk, mate rescue a workgroup per pair for every pair).
struct Knobs {
    bool timing = false, seed_one_base = false, dp_by_wave = false, dp_lane_always = false, dp_x1 = false, late_reseed = false, no_work_order = false, no_simple = false,
         simple_no_dp = false, cluster_by_lane = false, rescue_in_line = false, build_by_lane = false, no_sums_cache = false, prof_by_column = false,
         tier1_hist = false, dp_hist = false, no_tier_overlap = false, no_late_overlap = false, no_prof_overlap = false, no_prepack = false, no_tier1_grow = false;
    int seed_fm_budget = 6, build_wave_limit = 0x7fffffff;
    uint32_t order_min = 16384u;
};
static Knobs knobs_read()
{
    Knobs k;
    auto on = [](const char *name) { return getenv(name) != nullptr; };
    k.timing = on("MCX_TIMING"); k.seed_one_base = on("MCX_SEED_ONE_BASE"); k.dp_by_wave = on("MCX_DP_BY_WAVE"); k.dp_lane_always = on("MCX_DP_LANE_ALWAYS"); k.dp_x1 = on("MCX_DP_X1");
    k.late_reseed = on("MCX_LATE_RESEED"); k.no_work_order = on("MCX_NO_WORK_ORDER"); k.no_simple = on("MCX_NO_SIMPLE"); k.simple_no_dp = on("MCX_SIMPLE_NO_DP");
    k.cluster_by_lane = on("MCX_CLUSTER_BY_LANE"); k.rescue_in_line = on("MCX_RESCUE_IN_LINE"); k.build_by_lane = on("MCX_BUILD_BY_LANE");
    k.no_sums_cache = on("MCX_NO_SUMS_CACHE"); k.prof_by_column = on("MCX_PROF_BY_COLUMN"); k.tier1_hist = on("MCX_TIER1_HIST"); k.dp_hist = on("MCX_DP_HIST");
    k.no_tier_overlap = on("MCX_NO_TIER_OVERLAP"); k.no_late_overlap = on("MCX_NO_LATE_OVERLAP"); k.no_prof_overlap = on("MCX_NO_PROF_OVERLAP"); k.no_tier1_grow = on("MCX_NO_TIER1_GROW");
    // A batch packed on its way in (mcx_stream_submit_packed; MCX_PREPACK=1) pays on runtimes whose copies in and out overlap: 16.2-16.3 against 16.7-16.8 ms per
    // step on ROCm 7.2's; on one that puts both directions on one SDMA engine (HIP 7.0, what torch's wheel carries) the copy in ends late and the longer chain
    // behind it reaches into the next step: 17.5 against 17.1.  Off unless asked for: one fuzz round in 480 of the CLI with it on did not come out
    // identical to the oracle's (a difference or a failed command — the run kept only its count) and did not come back in 780 repeats; until that round is
    // understood the step packs its own reads.
    k.no_prepack = !on("MCX_PREPACK");
    if (const char *e = getenv("MCX_SEED_FM_BUDGET")) k.seed_fm_budget = std::max(1, atoi(e));
    if (const char *e = getenv("MCX_BUILD_WAVE_LIMIT")) k.build_wave_limit = atoi(e); // (tests: the bound sum from which k_build_wave hands a pair to one lane)
    if (const char *e = getenv("MCX_ORDER_MIN")) k.order_min = (uint32_t)std::max(1, atoi(e)); // (tests: small batches through k_simple and the order too)
    return k;
}

struct Tier {
    Caps caps;
    Layout lay;
    uint8_t *state = nullptr;
    uint32_t max_pairs = 0;
    uint32_t grow_to = 0; // the large tier: how many pair records it may grow to when a batch's heavy pairs do not fit one pass (tier1_grow; 0 = fixed)
};

constexpr uint32_t kPoutSel = 1u << 16; // pair outcomes gathered per copy (run_selection)
enum { CNT_TASKS = 0, CNT_RESCUE = 1 * kCntPad, CNT_JOB0 = 2 * kCntPad, CNT_JOB1 = 3 * kCntPad, CNT_JOB2 = 4 * kCntPad, CNT_JOB3 = 5 * kCntPad,
       CNT_JOB4 = 6 * kCntPad, CNT_JOB5 = 7 * kCntPad, CNT_OV = 8 * kCntPad, CNT_LF = 9 * kCntPad, CNT_CELLS = 10 * kCntPad, CNT_UNSUP = 11 * kCntPad,
       CNT_QUEUE = 12 * kCntPad, CNT_EARLY = 13 * kCntPad, CNT_LATE = 14 * kCntPad, CNT_RTASK = 15 * kCntPad, CNT_RPLAN = 16 * kCntPad, CNT_RESCUE_N = 17 * kCntPad, CNT_RSEED = 18 * kCntPad,
       CNT_SIMPLE = 19 * kCntPad, CNT_EARLY_HITS = 20 * kCntPad, CNT_SIMPLE_LATER = 21 * kCntPad, CNT_SIMPLE_JOBS = 22 * kCntPad, CNT_N = 23 * kCntPad,
       // behind the counters proper, cleared with them at the start of a pass (a memset in the middle of a pass was seen to sit 1.4 ms in its queue):
       CNT_ORDER = CNT_N, CNT_DP_SORT = CNT_ORDER + 16 * kCntPad, CNT_ALL = CNT_DP_SORT + 4 * kDpBuckets };
constexpr uint32_t kLateRoom = 2048; // pairs of a pass that may run over after clustering and still go through the large tier beside it

// What a pass over a selection of pairs works with besides the pair records: stream, counters, work lists, DP scratch.
// The context holds two sets, so that the large tier can map the heavy pairs of a pass (listed while the pass clusters)
// on a stream of its own while the rest of the pass is still under way.
struct PassRes {
    hipStream_t stream = nullptr;
    uint32_t *d_cnt = nullptr, *h_cnt = nullptr;
    uint2 *d_tasks = nullptr; uint32_t task_cap = 0;
    DpJob *d_jobs[kDpClasses] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; uint32_t job_cap[kDpClasses] = {0, 0, 0, 0, 0, 0};
    uint32_t *d_rescue = nullptr; uint32_t rescue_cap = 0;
    uint32_t *d_kscratch = nullptr; // k_rescue's scratch (not owned: a third of the context's)
    RescueTask *d_rtasks = nullptr; RescueRes *d_rres = nullptr; Hit *d_rseeds = nullptr; RescuePlan *d_rplans = nullptr; uint32_t *d_rescue_n = nullptr; // mate rescue window by window
    uint32_t rtask_cap = 0, rseed_cap = 0;
    uint8_t *d_dp_scratch[3] = {nullptr, nullptr, nullptr}; uint64_t dp_stride[3] = {0, 0, 0}; uint32_t dp_blocks[3] = {0, 0, 0};
    uint32_t *d_dp_lane = nullptr; uint32_t dp_lane_blocks = 0; // k_dp_lane's words for the three short lists (tiny | small | half), dp_lane_blocks wavefronts each
    uint32_t *d_dp_order[2] = {nullptr, nullptr}; // the two long lists by shape (k_dp_sort_*; their bucket counts and cursors: CNT_DP_SORT)
    hipStream_t dp_stream[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; hipEvent_t dp_fork = nullptr, dp_join[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t *d_ov = nullptr; uint32_t ov_cap = 0;
    uint32_t *d_sel_ids = nullptr; int32_t *d_est = nullptr;
    hipEvent_t ev[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

struct BatchRun { // the batch between mcx_batch_begin and mcx_batch_end
    bool open = false, sums_valid = false, keys_out = false;
    ReadBatch rb; int paired = 0;
    uint32_t n_pairs = 0, n_chunks = 0, longest = 0; // (longest read of the batch)
    AlnRec *recs = nullptr; uint32_t *cig = nullptr; // records [n_reads]; the batch's CIGAR pool
    uint32_t cig_cap = 0, cig_words = 0;             // its capacity (MCX_CIGAR_POOL_WORDS(n_reads)) and, once the batch is closed, the words taken
    int64_t read_base = 0, mapped = 0;
    unsigned long long hs[3] = {0, 0, 0};
    std::vector<uint32_t> ok, ds; // per chunk: proper pairs; summed distance, then summed read lengths
    const uint32_t *d_ok = nullptr, *d_ds = nullptr; // ... and where they lie on the device while sums_valid (the batch's tail keeps them in its own words, mcx_batch_sums in the per-read arrays)
    const uint64_t *d_sorted_keys = nullptr; uint64_t n_keys = 0; uint32_t n_sparse_keys = 0;
    std::chrono::steady_clock::time_point t0, t_begun; double ms_setup = 0; // (t_begun, ms_setup: MCX_TIMING)
    mcx_stats *stats = nullptr;
};

struct mcx_ctx {
    const mcx_index *idx = nullptr;
    bool counted = false; // (among idx->n_ctx)
    bool lens_checked = false; // the batch about to begin holds no read longer than max_read_len (mcx_stream_next says so for batches that came as 2-bit rows)
    const uint32_t *lens_checked_off = nullptr; const uint8_t *lens_checked_bases = nullptr; // ... said of THESE buffers (the slot's) and of no others
    // ... and is packed already (mcx_stream_submit_packed packed it behind its copy in, under the batch before it): where, from which bytes, mated or not, and its any-N word
    struct PrePacked { const uint32_t *packed = nullptr; const uint8_t *bases = nullptr; int paired = 0; const uint32_t *any_n = nullptr; } pre;
    int last_paired = 1;                 // what the last batch was mapped as: the guess a batch on its way in is packed under
    const uint32_t *packed_now = nullptr; // the 2-bit form the batch in flight is mapped from (d_packed, or a slot's)
    Knobs kn;
    Params pm;
    mcx_opts opts;
    hipStream_t stream = nullptr;
    Tier tier[2];
    uint64_t max_reads = 0, max_bases = 0;
    int rlen_max = 256;
    uint2 *d_tasks = nullptr; uint32_t task_cap = 0;
    DpJob *d_jobs[kDpClasses] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; uint32_t job_cap[kDpClasses] = {0, 0, 0, 0, 0, 0};
    uint32_t *d_cnt = nullptr;   // CNT_N counters
    uint32_t *h_cnt = nullptr;   // pinned mirror
    uint32_t *d_rescue = nullptr; uint32_t rescue_cap = 0;
    uint32_t *d_kscratch = nullptr; // k_rescue's scratch for reads with N
    RescueTask *d_rtasks = nullptr; RescueRes *d_rres = nullptr; Hit *d_rseeds = nullptr; RescuePlan *d_rplans = nullptr; uint32_t *d_rescue_n = nullptr;
    uint32_t rtask_cap = 0, rseed_cap = 0;
    hipEvent_t ev_pack[2] = {nullptr, nullptr};
    hipStream_t dp_stream[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; hipEvent_t dp_fork = nullptr, dp_join[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    uint8_t *d_dp_scratch[3] = {nullptr, nullptr, nullptr}; uint64_t dp_stride[3] = {0, 0, 0}; uint32_t dp_blocks[3] = {0, 0, 0};
    uint32_t *d_dp_lane = nullptr; uint32_t dp_lane_blocks = 0;
    uint32_t *d_dp_order[2] = {nullptr, nullptr};
    uint32_t *d_ov = nullptr; uint32_t ov_cap = 0;
    uint32_t *d_sel_ids = nullptr; int32_t *d_est = nullptr;
    uint32_t *d_read_ext = nullptr, *d_read_blocks = nullptr;
    uint32_t *d_packed = nullptr; int wpad = 0; // 2-bit form of the batch's reads
    uint32_t *d_order = nullptr; // the pairs of a pass by weight (k_order_*; their class counts: CNT_ORDER)
    uint8_t *d_done = nullptr;                           // per pair of a pass: k_simple wrote its records (the per-pair kernels skip it)
    uint32_t *d_sl_pairs = nullptr, *d_sl_list = nullptr; // the straight-line pairs that wait for a small gapped extension (SimpleLater)
    SimpleJob *d_sl_jobs = nullptr; SimpleRes *d_sl_res = nullptr; uint32_t sl_cap = 0;
    PairOut *d_pout = nullptr, *d_pout_sel = nullptr; // per-pair outcome of the finish stage; a gathered selection of it
    uint8_t *d_mapq = nullptr; int mapq_rows = 0;
    // -vcf bookkeeping (mcx_profile.h): caller-owned counter planes, per-read alignment detail
    uint32_t *prof_planes = nullptr; int prof_max_dup = 5, prof_max_clip = 5;
    ColItem *d_prof_items = nullptr; uint32_t prof_items_cap = 0; // fragments whose columns k_prof_cols walks
    uint16_t *d_prof_match = nullptr; bool prof_settled = false, prof_broken = false; // (broken: a settle failed half way — some planes scanned, some not)
    // exact-seed coverage as differences (mcx_profile.h); freed by mcx_profile_settle
    uint8_t *d_detail = nullptr; DetailLayout dlay;
    // One shard: a batch's bookkeeping is queued behind its mapping on a stream of its own and runs under the NEXT batch's kernels.  What the
    // mapping writes for it exists twice (detail records, flag bytes: the sets change places when a batch's bookkeeping is queued), the batch's reads are kept
    // in a copy of the context's (the caller's buffer is the caller's again when the call returns), and the bookkeeping has counters and an event list of its own.
    struct ProfLater {
        bool have = false, tried = false, pending = false, kept_now = false; // the resources exist; a batch's bookkeeping is queued and the host has not looked at its counts; the batch in flight has its reads kept
        hipStream_t stream = nullptr, keep_stream = nullptr; hipEvent_t go = nullptr, done = nullptr, kept = nullptr, begun = nullptr;
        uint8_t *d_detail_alt = nullptr, *d_admit_alt = nullptr, *d_keep_bases = nullptr, *d_keep_bases_alt = nullptr; uint32_t *d_keep_off = nullptr, *d_keep_off_alt = nullptr;
        uint32_t *d_cnt = nullptr, *h_cnt = nullptr; SparseRec *d_ev = nullptr; uint32_t ev_cap = 0;
        std::chrono::steady_clock::time_point t_queued;
    } later;
    uint64_t *d_keys[2] = {nullptr, nullptr}; uint8_t *d_admit = nullptr; void *d_sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
    SparseRec *d_sparse = nullptr; uint32_t sparse_cap = 0;
    struct Archive { SparseRec *d = nullptr; uint64_t n = 0, cap = 0; std::vector<mcx_sparse_rec> *host = nullptr; };
    Archive arch, arch_ev;                      // the tally records / discordant-pair events of the batches so far, still in HBM
    uint64_t arch_limit = (uint64_t)1 << 28;    // (at most 16 GB)
    SparseRec *h_sparse_pin = nullptr; uint32_t sparse_pin_recs = 1u << 18; // page-locked bounce buffer for their way to the host (16 MB)
    std::vector<mcx_sparse_rec> h_sparse, h_events; // tallies (followed by what the last mcx_profile_sparse* call appended for its caller: n_tally is where that starts); discordant-pair events ('E')
    size_t n_tally = 0;
    uint64_t keys_cap = 0;       // keys the sort buffers hold
    uint64_t *h_keys = nullptr; uint64_t h_keys_cap = 0; // pinned: the batch's keys for the exchange between shards
    // the tail of a whole-batch pass queued behind its kernels, before the host waits for them (mcx_map_batch_dev: queue_batch_tail) — the
    // seeding statistics, the per-chunk sums, the insert-size walk over them and the check of every pair's estimate, one copy back
    struct Tail {
        bool want = false, queued = false, ran = false; // asked for by the caller of this batch; queued by its pass and still standing; queued at all
        int64_t state0[3] = {1000, 0, 0};      // avgDist, pairs, distance sum before the batch
        uint32_t *d = nullptr;                 // device: [0..8) counters (n_redo, mapped), [8..8 + nc) the chunks' estimates
        uint32_t *h = nullptr; uint32_t cap = 0; // page-locked: counters[8] | flags[4] | statistics (3 x u64) | ok[nc] ds[nc] ls[nc] est[nc]
        hipEvent_t ev[2] = {nullptr, nullptr};
    } tail;
    uint32_t *d_batch_flags = nullptr; // [0] words taken in the batch's CIGAR pool, [1] longest read of the batch, [2] the pool ran over, [3] a read holds an N
    BatchRun run;
    PassRes t1;               // the large tier's own set (the members above are tier 0's); allocated when every suffix-array entry is resident
    bool overlap_tiers = false;
    bool dp_grown = false; // dp_scratch_grow() has had its one attempt
    uint32_t job2_seen = 0; // the longest 65-256-column DP list of a tier-0 pass so far
    volatile uint32_t *h_early = nullptr; uint32_t *d_early = nullptr; // page-locked words k_publish_early writes behind the clustering kernel: the host's view and the device's
    PassRes t2;               // a third set: the large tier's pass over the pairs that ran over after clustering (k_build's list)
    hipEvent_t ev_built = nullptr, ev_late_done = nullptr;
    bool overlap_late = false;
    hipEvent_t ev_clustered = nullptr;
    // mcx_stream_*: three batches in flight (copy in | kernels | copy out), each in a slot of its own
    struct Slot {
        uint8_t *d_bases = nullptr; uint32_t *d_off = nullptr; AlnRec *d_recs = nullptr; uint32_t *d_cig = nullptr;
        mcx_aln32 *d_recs32 = nullptr; // the records in 32 bytes each for their way out (mcx_stream_mapped32)
        uint32_t *d_codes = nullptr, *d_len = nullptr, *d_err = nullptr; uint64_t *d_odd = nullptr; uint32_t odd_cap = 0; // mcx_stream_submit_packed: what arrives; restored to d_bases / d_off
        uint32_t n_reads = 0; int state = 0; uint64_t seq = 0; // 0 free, 1 copy in started, 2 handed to the kernels, 3 copy out started
        bool lens_checked = false; // the batch came as 2-bit rows: no read is longer than the context's slots (k_unpack_reads / k_neutralize saw to it)
        uint32_t *h_err = nullptr; // pinned: d_err's word on its way out with the batch's records (mcx_stream_mapped / _mapped32 -> mcx_stream_collect)
        uint32_t *d_prepack = nullptr, *d_any_n = nullptr; bool prepacked = false; int pre_paired = 0; // k_pack_reads' output made on the way in
        hipEvent_t in_ready = nullptr, mapped = nullptr, out_done = nullptr;
    } slot[3];
    hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;
    uint64_t stream_seq = 0, stream_bytes_in = 0, stream_bytes_out = 0;
    void *d_scan_tmp = nullptr; size_t scan_tmp_bytes = 0; // the prefix sum of the read lengths (mcx_stream_submit_packed)
    void *files_state = nullptr; void (*files_drop)(void *) = nullptr; // mcx_files.cpp's batch buffers (mcx_ctx_files_slot)
    // staging for the host-buffer entry point
    uint8_t *d_bases = nullptr; uint32_t *d_off = nullptr; AlnRec *d_recs = nullptr; uint32_t *d_cig = nullptr;
    hipEvent_t ev[12];
};

extern "C" void mcx_opts_default(mcx_opts *o)
{
    o->alg = 0; o->max_pos_diff = 30; o->max_mismatch_rate = 0.05f; o->max_read_len = 256; o->max_batch_reads = 1 << 20;
}

static Caps tier0_caps()
{
    // (hit_seed above OCC_Thr: one seed at the occurrence limit plus the read's other seeds still fit)
    Caps c; c.hit_cap = 88; c.hit_seed = 56; c.cand_cap = 24; c.cand_seed = 12; // (rescue adds at most one candidate per candidate of the mate)
    c.frag_cap = 96; c.ops_cap = 2048; c.job_cap = 32;
    c.cig_cap = MCX_CIGAR_STRIDE; c.kmer_cap = 2048;
    return c;
}
static Caps tier1_caps(int rlen_max)
{
    Caps c;
    int seeds = rlen_max / (kMinSeedLength + 1) + 1;
    c.hit_cap = seeds * kOccThr + rlen_max / 8 + 16;
    c.cand_cap = c.hit_cap;
    c.hit_seed = c.hit_cap; c.cand_seed = c.cand_cap; // (hard bounds already count what the rescue can add)
    c.frag_cap = 7 * c.hit_cap / 2 + 16; // (k_build_wave places a candidate's fragments by their bound, 2 seeds + 2: a sixth more room than 3 per hit)
    c.ops_cap = 96 * 1024; c.job_cap = 2048;
    c.cig_cap = MCX_CIGAR_STRIDE; c.kmer_cap = 4096;
    return c;
}

// (MCX_TIMING) a host wait that took long says so
template <typename F>
static inline hipError_t timed_wait(bool on, const char *what, int line, F &&f)
{
    if (!on) return f();
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = f();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms > 15) fprintf(stderr, "[mcx] %.1f ms in %s (line %d)\n", ms, what, line);
    return e;
}
static std::atomic<size_t> g_dmalloc_bytes(0); // (MCX_TIMING: what a context takes)
template <class T>
static int dmalloc(T **p, size_t n, int line = __builtin_LINE())
{
    HIP_TRY(hipMalloc((void **)p, n * sizeof(T)));
    g_dmalloc_bytes += n * sizeof(T);
    static const bool log = getenv("MCX_ALLOC_LOG") != nullptr;
    if (log && n * sizeof(T) >= ((size_t)256 << 20)) fprintf(stderr, "[mcx alloc] %8.2f GB at mcx_pipeline.hip:%d\n", (double)(n * sizeof(T)) / 1e9, line);
    return 0;
}

static int ctx_fill(mcx_ctx *c, const mcx_index *idx, const mcx_opts &o);

extern "C" int mcx_ctx_create(const mcx_index *idx, const mcx_opts *opts, mcx_ctx **out)
{
    if (!idx || !out) return fail(MCX_ERR_ARG, "mcx_ctx_create: null argument");
    mcx_opts o;
    if (opts) o = *opts; else mcx_opts_default(&o);
    if (o.max_read_len < 32) o.max_read_len = 32;
    if (o.max_read_len > 1000) return fail(MCX_ERR_UNSUPPORTED, "max_read_len > 1000 is not supported");
    if (o.max_batch_reads < 2) o.max_batch_reads = 2;
    mcx_ctx *c = new mcx_ctx();
    c->kn = knobs_read();
    for (auto &e : c->ev) e = nullptr;
    const size_t before = g_dmalloc_bytes.load();
    const int rc = ctx_fill(c, idx, o);
    if (c->kn.timing) fprintf(stderr, "[mcx_ctx_create] %.2f GB of HBM for batches of %lld reads of up to %d bases\n", (double)(g_dmalloc_bytes.load() - before) / 1e9, (long long)o.max_batch_reads, (int)o.max_read_len);
    if (rc) { mcx_ctx_free(c); return rc; } // (every pointer of the context starts null: a caller that retries with a smaller batch finds the HBM free again)
    idx->n_ctx++; c->counted = true;
    *out = c;
    return 0;
}

struct mcx_ctx;
// the window lists of mate rescue for a set of pass resources that lists up to `pairs` pairs
static int rescue_alloc(mcx_ctx *c, uint64_t pairs, bool large, RescueTask **tasks, RescueRes **res, Hit **seeds, RescuePlan **plans, uint32_t **ids_n, uint32_t *task_cap, uint32_t *seed_cap)
{
    int rc;
    // (the large tier's pairs have hundreds of candidates: many more windows per pair than tier 0's, whose lists stay short)
    *task_cap = large ? 1u << 22 : (uint32_t)std::min<uint64_t>(std::max<uint64_t>(pairs, 1u << 18), 1u << 21);
    if (const char *e = getenv("MCX_RESCUE_TASK_CAP")) *task_cap = (uint32_t)std::max(16, atoi(e)); // (tests: make the list run over)
    *seed_cap = large ? *task_cap * 4 : *task_cap; // seeds are kept only of windows that beat the mate's best candidate (a few per pair; a heavy pair's windows mostly do)
    if ((rc = dmalloc(tasks, *task_cap))) return rc;
    if ((rc = dmalloc(res, *task_cap))) return rc;
    if ((rc = dmalloc(seeds, (size_t)*seed_cap))) return rc;
    if ((rc = dmalloc(plans, pairs))) return rc;
    if ((rc = dmalloc(ids_n, pairs))) return rc;
    return 0;
}

// a set of pass resources beside the context's own (PassRes): work lists for `pairs` pairs at a time, selections of up to `sel_cap`
static int passres_alloc(mcx_ctx *c, PassRes &t, uint64_t pairs, uint64_t sel_cap, int priority)
{
    int rc;
    HIP_TRY(hipStreamCreateWithPriority(&t.stream, hipStreamNonBlocking, priority));
    // side streams for the DP lists only where lists are long enough to share the chip: a stream that exists lands on one of the
    // runtime's few hardware queues, and a queue that waits for an event holds up every stream folded onto it (the late pairs' pass
    // once sat 4 ms behind the large tier's DP fork that way)
    const int n_side = pairs >= 4096 ? 2 : 0;
    for (int k = 0; k < n_side; k++) { HIP_TRY(hipStreamCreateWithPriority(&t.dp_stream[k], hipStreamNonBlocking, priority)); HIP_TRY(hipEventCreateWithFlags(&t.dp_join[k], hipEventDisableTiming)); }
    HIP_TRY(hipEventCreateWithFlags(&t.dp_fork, hipEventDisableTiming));
    for (auto &e : t.ev) HIP_TRY(hipEventCreate(&e));
    if ((rc = dmalloc(&t.d_cnt, CNT_ALL))) return rc;
    HIP_TRY(hipHostMalloc((void **)&t.h_cnt, CNT_N * sizeof(uint32_t)));
    for (int k = 0; k < kDpClasses; k++) {
        t.job_cap[k] = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(pairs * 16, 1u << 20), c->job_cap[k]);
        if ((rc = dmalloc(&t.d_jobs[k], t.job_cap[k]))) return rc;
    }
    t.rescue_cap = (uint32_t)pairs;
    if ((rc = dmalloc(&t.d_rescue, t.rescue_cap))) return rc;
    if ((rc = rescue_alloc(c, pairs, pairs >= 4096, &t.d_rtasks, &t.d_rres, &t.d_rseeds, &t.d_rplans, &t.d_rescue_n, &t.rtask_cap, &t.rseed_cap))) return rc;
    const uint32_t blocks1[3] = {pairs >= 4096 ? 2048u : 256u, pairs >= 4096 ? 2048u : 128u, 256};
    for (int k = 0; k < 3; k++) {
        t.dp_stride[k] = c->dp_stride[k]; t.dp_blocks[k] = blocks1[k];
        if ((rc = dmalloc(&t.d_dp_scratch[k], (size_t)t.dp_stride[k] * t.dp_blocks[k]))) return rc;
    }
    t.dp_lane_blocks = pairs >= 4096 ? 2048u : 256u;
    if ((rc = dmalloc(&t.d_dp_lane, (size_t)(lane_short_words(0) + lane_short_words(1) + lane_short_words(2)) * t.dp_lane_blocks))) return rc;
    for (int k = 0; k < 2; k++) if ((rc = dmalloc(&t.d_dp_order[k], t.job_cap[1 + k]))) return rc;
    t.ov_cap = (uint32_t)sel_cap;
    if ((rc = dmalloc(&t.d_ov, t.ov_cap))) return rc;
    if ((rc = dmalloc(&t.d_sel_ids, sel_cap))) return rc;
    if ((rc = dmalloc(&t.d_est, sel_cap))) return rc;
    return 0;
}

static void passres_free(PassRes &t)
{
    void *q[] = {t.d_cnt, t.d_jobs[0], t.d_jobs[1], t.d_jobs[2], t.d_jobs[3], t.d_jobs[4], t.d_jobs[5], t.d_rescue, t.d_dp_scratch[0], t.d_dp_scratch[1],
                 t.d_dp_scratch[2], t.d_ov, t.d_sel_ids, t.d_est, t.d_rtasks, t.d_rres, t.d_rseeds, t.d_rplans, t.d_rescue_n, t.d_dp_lane, t.d_dp_order[0], t.d_dp_order[1]};
    for (void *x : q) if (x) (void)hipFree(x);
    if (t.h_cnt) (void)hipHostFree(t.h_cnt);
    for (auto &e : t.ev) if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < 5; k++) { if (t.dp_stream[k]) (void)hipStreamDestroy(t.dp_stream[k]); if (t.dp_join[k]) (void)hipEventDestroy(t.dp_join[k]); }
    if (t.dp_fork) (void)hipEventDestroy(t.dp_fork);
    if (t.stream) (void)hipStreamDestroy(t.stream);
}

static int ctx_fill(mcx_ctx *c, const mcx_index *idx, const mcx_opts &o)
{
    c->idx = idx; c->opts = o;
    c->pm.max_pos_diff = o.max_pos_diff; c->pm.max_mm_rate = o.max_mismatch_rate; c->pm.use_nw = o.alg == 0; c->pm.paired = 1;
    c->rlen_max = o.max_read_len;
    c->max_reads = (uint64_t)o.max_batch_reads;
    c->max_bases = c->max_reads * (uint64_t)c->rlen_max;
    HIP_TRY(hipSetDevice(idx->device));
    HIP_TRY(hipStreamCreate(&c->stream));
    for (int k = 0; k < 2; k++) { HIP_TRY(hipStreamCreateWithFlags(&c->dp_stream[k], hipStreamNonBlocking)); HIP_TRY(hipEventCreateWithFlags(&c->dp_join[k], hipEventDisableTiming)); }
    HIP_TRY(hipEventCreateWithFlags(&c->dp_fork, hipEventDisableTiming));
    for (auto &e : c->ev_pack) HIP_TRY(hipEventCreate(&e));
    for (auto &e : c->ev) HIP_TRY(hipEventCreate(&e));
    int rc = 0;
    c->tier[0].caps = tier0_caps(); c->tier[0].lay = make_layout(c->tier[0].caps); c->tier[0].max_pairs = (uint32_t)c->max_reads;
    c->tier[1].caps = tier1_caps(c->rlen_max); c->tier[1].lay = make_layout(c->tier[1].caps);
    // (the heavy pairs of a batch in as few passes as possible — a pass is bound by its slowest pair, not by its size, and passes follow
    //  one another: BASELINE config 5 sends 4.5 % of its pairs here, five passes of 36 k pairs with 8 GB of records — with room in
    //  proportion to the batch: 3 KB per read, 2-24 GB (MCX_TIER1_GB overrides))
    uint64_t t1_bytes = std::min<uint64_t>(std::max<uint64_t>(c->max_reads * 3072, (uint64_t)2 << 30), (uint64_t)24 << 30);
    const uint64_t t1_limit = std::min<uint64_t>(c->max_reads, 262144);
    if (const char *e = getenv("MCX_TIER1_GB")) t1_bytes = (uint64_t)std::max(1, atoi(e)) << 30;
    else if (const char *e2 = getenv("MCX_TIER1_START_GB")) t1_bytes = (uint64_t)(std::max(0.05, atof(e2)) * (double)(1 << 30)); // (tests: a small start that may grow)
    c->tier[1].max_pairs = (uint32_t)std::max<uint64_t>(1024, std::min<uint64_t>(t1_limit, t1_bytes / (uint64_t)c->tier[1].lay.stride));
    // (a batch whose heavy pairs do not fit these records in one pass makes them grow, HBM permitting: tier1_grow().  A size that was asked for stays)
    c->tier[1].grow_to = getenv("MCX_TIER1_GB") || c->kn.no_tier1_grow ? 0u : (uint32_t)std::max<uint64_t>(t1_limit, c->tier[1].max_pairs);
    // (tier 0's records are allocated by the first batch: a paired batch of max_reads reads is max_reads / 2 pairs, and at 8 KB a
    //  record the other half is 33 GB at 8 M reads — only single-end batches need a record per read)
    c->tier[0].max_pairs = 0;
    if ((rc = dmalloc(&c->tier[1].state, (size_t)c->tier[1].lay.stride * c->tier[1].max_pairs))) return rc;
    c->task_cap = (uint32_t)std::min<uint64_t>(c->max_reads * 24, 0x7fffffffu);
    if ((rc = dmalloc(&c->d_tasks, c->task_cap))) return rc;
    for (int k = 0; k < kDpClasses; k++) {
        c->job_cap[k] = (uint32_t)std::min<uint64_t>(c->max_reads * ((k == 0 || k == 4) ? 4 : ((k == 1 || k == 5) ? 2 : 1)) + 1024, 0x7fffffffu);
        if (const char *e = getenv("MCX_JOB_CAP")) c->job_cap[k] = std::min<uint32_t>(c->job_cap[k], (uint32_t)std::max(1024, atoi(e))); // (tests: make the lists run over)
        if ((rc = dmalloc(&c->d_jobs[k], c->job_cap[k]))) return rc;
    }
    if ((rc = dmalloc(&c->d_cnt, CNT_ALL))) return rc;
    HIP_TRY(hipHostMalloc((void **)&c->h_cnt, (CNT_N + 8) * sizeof(uint32_t))); // (+8: the seeding statistics of a batch, on their way to batch_close)
    c->rescue_cap = (uint32_t)c->max_reads;
    if ((rc = dmalloc(&c->d_rescue, c->rescue_cap))) return rc;
    if ((rc = rescue_alloc(c, c->max_reads, false, &c->d_rtasks, &c->d_rres, &c->d_rseeds, &c->d_rplans, &c->d_rescue_n, &c->rtask_cap, &c->rseed_cap))) return rc;
    if ((rc = dmalloc(&c->d_kscratch, 3 * (size_t)kRescueBlocks * kRescueScratchWords))) return rc; // (one part per set of pass resources)
    // DP traceback spill per block: 4 KB of sequences + (qlen + tlen - 1) * tlen direction bytes
    // (the two grouped classes: a wavefront's stretch holds the strings and traceback bytes of a group of problems — k_dp_group —,
    //  at least those of the class's largest one)
    const uint64_t spill[3] = {(uint64_t)512 << 10, (uint64_t)1 << 20, kDpSpillSeq + (uint64_t)(2048 + 1024) * 1024};
    uint32_t blocks[3] = {8192, 4096, 512}; // (the 65-256-column class at 2048 or 8192 wavefronts: the same DP stage, 3.3-3.8 ms)
    if (const char *e = getenv("MCX_DP_BLOCKS1")) blocks[1] = (uint32_t)std::max(256, atoi(e)); // (experiments; a size that was asked for stays: dp_scratch_grow)
    for (int k = 0; k < 3; k++) {
        c->dp_stride[k] = spill[k]; c->dp_blocks[k] = blocks[k];
        if ((rc = dmalloc(&c->d_dp_scratch[k], (size_t)spill[k] * blocks[k]))) return rc;
    }
    c->dp_lane_blocks = 4096;
    if ((rc = dmalloc(&c->d_dp_lane, (size_t)(lane_short_words(0) + lane_short_words(1) + lane_short_words(2)) * c->dp_lane_blocks))) return rc;
    for (int k = 0; k < 2; k++) if ((rc = dmalloc(&c->d_dp_order[k], c->job_cap[1 + k]))) return rc;
    c->ov_cap = (uint32_t)c->max_reads;
    if ((rc = dmalloc(&c->d_ov, c->ov_cap))) return rc;
    if ((rc = dmalloc(&c->d_sel_ids, c->max_reads))) return rc;
    if ((rc = dmalloc(&c->d_est, c->max_reads))) return rc;
    if ((rc = dmalloc(&c->d_read_ext, c->max_reads))) return rc;
    if ((rc = dmalloc(&c->d_read_blocks, c->max_reads))) return rc;
    if ((rc = dmalloc(&c->d_batch_flags, 4))) return rc;
    HIP_TRY(hipMemset(c->d_batch_flags, 0, 4 * sizeof(uint32_t)));
    c->wpad = (packed_words(c->rlen_max) + 3) & ~3;
    if ((rc = dmalloc(&c->d_packed, c->max_reads * (uint64_t)c->wpad))) return rc;
    if ((rc = dmalloc(&c->d_pout, c->max_reads))) return rc;
    if ((rc = dmalloc(&c->d_pout_sel, kPoutSel))) return rc;
    if ((rc = dmalloc(&c->d_order, c->max_reads))) return rc;
    if ((rc = dmalloc(&c->d_done, c->max_reads))) return rc;
    c->sl_cap = (uint32_t)std::max<uint64_t>(c->max_reads / 4, 1024);
    if ((rc = dmalloc(&c->d_sl_pairs, c->sl_cap))) return rc;
    if ((rc = dmalloc(&c->d_sl_list, (uint64_t)c->sl_cap * kSimpleJobs))) return rc;
    if ((rc = dmalloc(&c->d_sl_jobs, (uint64_t)c->sl_cap * kSimpleJobs))) return rc;
    if ((rc = dmalloc(&c->d_sl_res, (uint64_t)c->sl_cap * kSimpleJobs))) return rc;
    // EvaluateMAPQ (SamReport.cpp:86-101) tabulated on the host so that the double-precision
    // log() is the host libm's, exactly as in the reference
    c->mapq_rows = c->rlen_max + 64;
    std::vector<uint8_t> tab((size_t)c->mapq_rows * 6, 0);
    for (int s = 1; s < c->mapq_rows; s++)
        for (int d = 1; d <= 5 && d < s; d++) {
            int sub = s - d;
            int q = (int)(30 * (1 - (float)(s - sub) / s) * log(s) + 0.4999);
            tab[(size_t)s * 6 + d] = (uint8_t)(q > 60 ? 60 : q);
        }
    if ((rc = dmalloc(&c->d_mapq, tab.size()))) return rc;
    HIP_TRY(hipMemcpy(c->d_mapq, tab.data(), tab.size(), hipMemcpyHostToDevice));
    // the large tier's own stream, counters and lists: with them it maps the heavy pairs of a pass while the pass goes on
    // (without the full suffix array it would also need an SA task list of its own: then the tiers run one after the other)
    if (idx->view.sa_full && !c->kn.no_tier_overlap) {
        // its kernels are as long as their slowest pair, and the batch waits for them: their waves go first
        // (human-like bench genome: 40.0 -> 37.4 ms per step)
        int pr_lo = 0, pr_hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi));
        if ((rc = passres_alloc(c, c->t1, std::max(c->tier[1].max_pairs, c->tier[1].grow_to), c->max_reads, pr_hi))) return rc; // (lists for as many pairs as the records may grow to)
        c->t1.d_kscratch = c->d_kscratch + (size_t)kRescueBlocks * kRescueScratchWords;
        HIP_TRY(hipEventCreate(&c->ev_clustered));
        c->overlap_tiers = true;
        {
            void *h = nullptr, *d = nullptr;
            if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) { c->h_early = (volatile uint32_t *)h; c->d_early = (uint32_t *)d; }
            else { (void)hipGetLastError(); if (h) (void)hipHostFree(h); } // (without it: the copies behind the kernel, as before)
        }
        // and a small third set for the pairs that run over after clustering: they go through the large tier while the pass's
        // DP and finish stages run, in the last kLateRoom records of the tier, instead of in a pass of their own after it
        if (c->tier[1].max_pairs >= 4 * kLateRoom && !c->kn.no_late_overlap) {
            if ((rc = passres_alloc(c, c->t2, kLateRoom, kLateRoom, pr_hi))) return rc;
            c->t2.d_kscratch = c->d_kscratch + 2 * (size_t)kRescueBlocks * kRescueScratchWords;
            HIP_TRY(hipEventCreateWithFlags(&c->ev_built, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->ev_late_done, hipEventDisableTiming));
            c->overlap_late = true;
        }
    }
    return 0;
}

extern "C" void mcx_ctx_free(mcx_ctx *c)
{
    if (!c) return;
    if (c->counted && c->idx && --c->idx->n_ctx == 0 && c->idx->orphan.load()) delete c->idx; // (the index was freed first: its host object waited for this)
    if (c->files_state && c->files_drop) c->files_drop(c->files_state);
    void *p[] = {c->tier[0].state, c->tier[1].state, c->d_tasks, c->d_jobs[0], c->d_jobs[1], c->d_jobs[2], c->d_jobs[3], c->d_jobs[4], c->d_jobs[5],
                 c->d_cnt, c->d_rescue, c->d_kscratch, c->d_rtasks, c->d_rres, c->d_rseeds, c->d_rplans, c->d_rescue_n, c->d_dp_scratch[0], c->d_dp_scratch[1], c->d_dp_scratch[2],
                 c->d_ov, c->d_sel_ids, c->d_est, c->d_read_ext, c->d_read_blocks, c->d_pout, c->d_mapq,
                 c->d_dp_lane, c->d_dp_order[0], c->d_dp_order[1], c->d_bases, c->d_off, c->d_recs, c->d_cig, c->d_detail, c->d_keys[0], c->d_keys[1], c->d_admit, c->d_sort_tmp, c->d_sparse, c->d_pout_sel, c->d_order, c->d_done, c->d_sl_pairs, c->d_sl_list, c->d_sl_jobs, c->d_sl_res, c->d_packed, c->d_batch_flags, c->d_scan_tmp, c->d_prof_match, c->d_prof_items};
    for (void *q : p) if (q) (void)hipFree(q);
    if (c->h_cnt) (void)hipHostFree(c->h_cnt);
    if (c->h_keys) (void)hipHostFree(c->h_keys);
    if (c->h_sparse_pin) (void)hipHostFree(c->h_sparse_pin);
    if (c->arch.d) (void)hipFree(c->arch.d);
    if (c->arch_ev.d) (void)hipFree(c->arch_ev.d);
    passres_free(c->t1); passres_free(c->t2);
    for (hipEvent_t e : {c->ev_clustered, c->ev_built, c->ev_late_done, c->tail.ev[0], c->tail.ev[1]}) if (e) (void)hipEventDestroy(e);
    {
        auto &L = c->later;
        if (L.stream) (void)hipStreamSynchronize(L.stream);
        for (void *q : {(void *)L.d_detail_alt, (void *)L.d_admit_alt, (void *)L.d_keep_bases, (void *)L.d_keep_off, (void *)L.d_keep_bases_alt, (void *)L.d_keep_off_alt, (void *)L.d_cnt, (void *)L.d_ev}) if (q) (void)hipFree(q);
        if (L.h_cnt) (void)hipHostFree(L.h_cnt);
        for (hipEvent_t e : {L.go, L.done, L.kept, L.begun}) if (e) (void)hipEventDestroy(e);
        if (L.stream) (void)hipStreamDestroy(L.stream);
        if (L.keep_stream) (void)hipStreamDestroy(L.keep_stream);
    }
    if (c->h_early) (void)hipHostFree((void *)c->h_early);
    if (c->tail.d) (void)hipFree(c->tail.d);
    if (c->tail.h) (void)hipHostFree(c->tail.h);
    for (auto &sl : c->slot) {
        void *q[] = {sl.d_bases, sl.d_off, sl.d_recs, sl.d_cig, sl.d_codes, sl.d_len, sl.d_odd, sl.d_err, sl.d_recs32, sl.d_prepack, sl.d_any_n};
        for (void *x : q) if (x) (void)hipFree(x);
        if (sl.h_err) (void)hipHostFree(sl.h_err);
        for (hipEvent_t e : {sl.in_ready, sl.mapped, sl.out_done}) if (e) (void)hipEventDestroy(e);
    }
    if (c->h2d_stream) (void)hipStreamDestroy(c->h2d_stream);
    if (c->d2h_stream) (void)hipStreamDestroy(c->d2h_stream);
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (int k = 0; k < 5; k++) { if (c->dp_stream[k]) (void)hipStreamDestroy(c->dp_stream[k]); if (c->dp_join[k]) (void)hipEventDestroy(c->dp_join[k]); }
    if (c->dp_fork) (void)hipEventDestroy(c->dp_fork);
    for (auto &e : c->ev_pack) if (e) (void)hipEventDestroy(e);
    delete c;
}

static Ctx make_ctx(const mcx_ctx *c, int tier, int paired)
{
    Ctx cx;
    cx.ix = c->idx->view; cx.pm = c->pm; cx.pm.paired = paired;
    if (c->kn.seed_one_base) cx.ix.rank2 = nullptr; // (experiments, tests: the walk one base per step although the pair records are there)
    cx.caps = c->tier[tier].caps; cx.lay = c->tier[tier].lay; cx.state = c->tier[tier].state;
    cx.mapq_tab = c->d_mapq; cx.mapq_rows = c->mapq_rows; cx.dp_summary = 1;
    cx.detail = c->prof_planes ? c->d_detail : nullptr; cx.dlay = c->dlay;
    cx.cig_pool = c->run.cig; cx.cig_pool_n = c->d_batch_flags; cx.cig_pool_cap = c->run.cig_cap;
    cx.packed = c->packed_now ? c->packed_now : c->d_packed; cx.wpad = c->wpad; cx.read_ext = c->d_read_ext; cx.seed_pool = nullptr;
    return cx;
}

// ---------------------------------------------------------------------------------------------
// one tier over a selection of pairs
// ---------------------------------------------------------------------------------------------
struct StageMs { float seed, sa, cluster, rescue, build, dp, finish; };

constexpr int kListOverflow = 1; // (internal) a work list of run_pairs was too short for the selection

// tier 0's set: the context's own members
static PassRes res_tier0(mcx_ctx *c)
{
    PassRes r;
    r.stream = c->stream; r.d_cnt = c->d_cnt; r.h_cnt = c->h_cnt; r.d_tasks = c->d_tasks; r.task_cap = c->task_cap;
    for (int k = 0; k < kDpClasses; k++) { r.d_jobs[k] = c->d_jobs[k]; r.job_cap[k] = c->job_cap[k]; }
    r.d_rescue = c->d_rescue; r.rescue_cap = c->rescue_cap; r.d_kscratch = c->d_kscratch;
    r.d_rtasks = c->d_rtasks; r.d_rres = c->d_rres; r.d_rseeds = c->d_rseeds; r.d_rplans = c->d_rplans; r.d_rescue_n = c->d_rescue_n; r.rtask_cap = c->rtask_cap; r.rseed_cap = c->rseed_cap;
    for (int k = 0; k < 3; k++) { r.d_dp_scratch[k] = c->d_dp_scratch[k]; r.dp_stride[k] = c->dp_stride[k]; r.dp_blocks[k] = c->dp_blocks[k]; }
    r.d_dp_lane = c->d_dp_lane; r.dp_lane_blocks = c->dp_lane_blocks; r.d_dp_order[0] = c->d_dp_order[0]; r.d_dp_order[1] = c->d_dp_order[1];
    for (int k = 0; k < 5; k++) { r.dp_stream[k] = c->dp_stream[k]; r.dp_join[k] = c->dp_join[k]; }
    r.dp_fork = c->dp_fork; r.d_ov = c->d_ov; r.ov_cap = c->ov_cap; r.d_sel_ids = c->d_sel_ids; r.d_est = c->d_est;
    for (int k = 0; k < 12; k++) r.ev[k] = c->ev[k];
    return r;
}

// the DP job lists of a pass, one kernel per size class: they work on disjoint lists and are each bound by latency at
// modest occupancy, so side streams let them share the chip instead of queueing behind one another
// A long list is worth a lane per problem; a short one (the large tier's, a replay's, the late pairs': a few thousand problems of
// 100 x 100 cells) is done sooner with a wavefront per problem — fewer problems than the chip has lanes, each 60 times quicker
// that way.  The list's length is known on the device only, so both kernels are launched and the one whose turn it is not leaves at
// once: k_dp_group below kDpLaneMin problems, k_dp_lane from there on.
constexpr uint32_t kDpLaneMin[2] = {65536, 131072}; // targets of 17-64 bases (mean 45 x 45 cells), of 65-256 (95 x 95): two wavefronts per SIMD's worth of problems

// The large tier's records grow when a batch sends it more pairs than one pass holds.  Its passes follow one another, each as long as its slowest pair,
// and only the first runs beside tier 0: BASELINE config 5 (185 k heavy pairs of 4 M, 260 KB of records each) took two passes of 99 k and 86 k with the
// 24 GB the context starts with, the second one alone on the chip for 17 ms after tier 0 had finished — one pass of 185 k: 117.2 -> 108.5 ms a step.
// Called between passes (nothing of the tier is under way: every earlier pass was waited for by tier1_pass_end), with the batch's tier-0 kernels queued:
// hipFree waits for the device, once or twice in a run.  Grows only into HBM that is free beyond kGrowKeep, so that what a caller allocates later (a
// profile's planes are attached before the first batch; a second set of detail records is not) still finds room; never shrinks.
constexpr uint64_t kGrowKeep = (uint64_t)16 << 30;
static int tier1_grow(mcx_ctx *c, uint64_t pairs_wanted)
{
    Tier &t = c->tier[1];
    if (!t.grow_to || pairs_wanted <= t.max_pairs || t.max_pairs >= t.grow_to) return 0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (const char *e = getenv("MCX_HBM_CAP_GB")) { // (tests: as if the device had this much)
        const uint64_t cap = (uint64_t)std::max(1, atoi(e)) << 30, used = total_b - free_b;
        free_b = cap > used ? cap - used : 0;
    }
    const uint64_t have = (uint64_t)t.lay.stride * t.max_pairs;
    uint64_t want = std::min<uint64_t>(t.grow_to, pairs_wanted + pairs_wanted / 8 + 1024); // (an eighth more: the next batch's count differs a little)
    const uint64_t room = free_b + have > kGrowKeep ? (free_b + have - kGrowKeep) / (uint64_t)t.lay.stride : 0;
    want = std::min(want, room);
    if (want < pairs_wanted || want <= t.max_pairs) return 0; // (not in one pass anyway: what is there stays)
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(t.state)); t.state = nullptr;
    uint8_t *p = nullptr;
    if (hipMalloc((void **)&p, (size_t)t.lay.stride * want) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc((void **)&p, (size_t)have) != hipSuccess) { (void)hipGetLastError(); t.max_pairs = 0; return fail(MCX_ERR_DEVICE, "the large tier's records: out of device memory"); }
        t.state = p; t.grow_to = 0;
        return 0;
    }
    if (c->kn.timing || getenv("MCX_ALLOC_LOG")) fprintf(stderr, "[mcx] the large tier's records grow from %u to %llu pairs (%.1f -> %.1f GB): %llu heavy pairs in this batch\n",
                                                       t.max_pairs, (unsigned long long)want, (double)have / 1e9, (double)t.lay.stride * want / 1e9, (unsigned long long)pairs_wanted);
    g_dmalloc_bytes += (size_t)t.lay.stride * want - have;
    t.state = p; t.max_pairs = (uint32_t)want;
    return 0;
}

// The scratch of the 65-256-column list grows when a batch fills it.  A wavefront of k_dp_lane2 keeps the traceback bits of its 128 problems in a stretch
// sized for the list's largest shape (2.2 MB at 250 bp), and the 4 GB a context starts with hold 1920 of them — fewer than two per SIMD, and this list's
// kernel is the DP stage's last to end at config 5 (1.3 M problems: 106.8 / 109.6 ms a step with 4 GB, 100.8 / 102.9 with 8).  Called from batch_close()
// with the batch's own kernels through; into HBM that is free beyond kGrowKeep, at most 12 GB, once.
static int dp_scratch_grow(mcx_ctx *c, uint32_t list_len)
{
    if (c->dp_grown || getenv("MCX_DP_BLOCKS1") || c->kn.no_tier1_grow) return 0;
    const bool nw = c->pm.use_nw != 0;
    const uint64_t w2 = lane_stride_words<16>(nw, c->rlen_max, 16, true) * 4; // bytes a wavefront's stretch takes
    const uint64_t have = c->dp_stride[1] * c->dp_blocks[1];
    uint64_t enough = 2 * 128 * (have / w2); // (fewer than two groups per wavefront there is room for: short enough)
    if (const char *e = getenv("MCX_DP_GROW_MIN")) enough = (uint64_t)std::max(0, atoi(e)); // (tests)
    if ((uint64_t)list_len <= enough) return 0;
    c->dp_grown = true; // (one attempt)
    const uint64_t want_bytes = std::min<uint64_t>(4096 * w2, (uint64_t)12 << 30);
    const uint32_t want = (uint32_t)((want_bytes + c->dp_stride[1] - 1) / c->dp_stride[1]);
    if (want <= c->dp_blocks[1]) return 0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (const char *e = getenv("MCX_HBM_CAP_GB")) { const uint64_t cap = (uint64_t)std::max(1, atoi(e)) << 30, used = total_b - free_b; free_b = cap > used ? cap - used : 0; }
    if (free_b + have < (uint64_t)want * c->dp_stride[1] + kGrowKeep) return 0;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(c->d_dp_scratch[1])); c->d_dp_scratch[1] = nullptr;
    uint8_t *p = nullptr;
    uint32_t got = want;
    if (hipMalloc((void **)&p, (size_t)want * c->dp_stride[1]) != hipSuccess) {
        (void)hipGetLastError();
        got = c->dp_blocks[1];
        if (hipMalloc((void **)&p, (size_t)have) != hipSuccess) { (void)hipGetLastError(); return fail(MCX_ERR_DEVICE, "the DP lists' scratch: out of device memory"); }
    }
    if (got != c->dp_blocks[1] && (c->kn.timing || getenv("MCX_ALLOC_LOG")))
        fprintf(stderr, "[mcx] the 65-256-column DP list's scratch grows from %.1f to %.1f GB: %u problems in this batch\n", (double)have / 1e9, (double)got * c->dp_stride[1] / 1e9, list_len);
    g_dmalloc_bytes += (size_t)got * c->dp_stride[1] - have;
    c->d_dp_scratch[1] = p; c->dp_blocks[1] = got;
    return 0;
}

static int launch_dp(const Knobs &kn, const PassRes &R, const Ctx &cx, const JobSinks &sinks, const ReadBatch &rb, const PairSel &sel, int rlen_max)
{
    hipStream_t s = R.stream;
    // three chains of about the same length (the runtime folds streams onto a few hardware queues anyway: more streams only
    // make the pairing of kernels on a queue a matter of luck); a set of pass resources without side streams runs them in turn
    const int n_side = R.dp_stream[1] ? 2 : 0;
    hipStream_t side0 = n_side ? R.dp_stream[0] : s, side1 = n_side ? R.dp_stream[1] : s;
    HIP_TRY(hipEventRecord(R.dp_fork, s));
    for (int k = 0; k < n_side; k++) HIP_TRY(hipStreamWaitEvent(R.dp_stream[k], R.dp_fork, 0));
    const bool by_wave = kn.dp_by_wave; // (experiments, and the A/B of the parity tests: the wavefront-per-problem kernels of mcx_dp.h)
    if (!by_wave) {
        // every list but the largest problems': one problem per lane (mcx_dp_lane.h).  A list's stretch of scratch per wavefront is
        // sized for its largest possible group; the two long lists share the wavefront kernels' buffers
        const bool nw = cx.pm.use_nw != 0;
        uint32_t *unsup = sinks.unsupported;
        const bool always = kn.dp_lane_always;
        uint32_t lane_min[2] = {always ? 0u : kDpLaneMin[0], always ? 0u : kDpLaneMin[1]};
        // two problems per lane in 16-bit halves (k_dp_lane2) wherever the scores fit them with room to spare: queries + targets far below kNeg2's reach, and a cell's
        // s~ (never below -2 (i + j) - 2: mismatches down the diagonal and one gap) within the fourteen bits a strip's edge word keeps of it
        const bool x2 = !kn.dp_x1 && rlen_max + 256 <= 3000;
        const uint64_t w1 = lane_stride_words<16>(nw, rlen_max, 4, x2), w2 = lane_stride_words<16>(nw, rlen_max, 16, x2);
        const unsigned b1 = (unsigned)std::min<uint64_t>(4096, R.dp_stride[0] * R.dp_blocks[0] / (w1 * 4)), b2 = (unsigned)std::min<uint64_t>(4096, R.dp_stride[1] * R.dp_blocks[1] / (w2 * 4));
        // (a set of pass resources whose scratch does not hold one lane group for reads this long — the small sets with a large max_read_len —
        //  leaves that list to the wavefront kernel whatever its length)
        if (b1 == 0) lane_min[0] = 0xFFFFFFFFu;
        if (b2 == 0) lane_min[1] = 0xFFFFFFFFu;
        const bool by_shape = true;
        const uint32_t *ord[2] = {nullptr, nullptr};
        hipStream_t st[2] = {s, side1};
        if (by_shape) {
            const int row_shift = rlen_max <= 256 ? 2 : (rlen_max <= 512 ? 3 : (rlen_max <= 1024 ? 4 : 6)); // (64 row classes cover the longest query)
            for (int k = 0; k < 2; k++) {
                if (lane_min[k] == 0xFFFFFFFFu) continue;
                uint32_t *counts = R.d_cnt + CNT_DP_SORT + 2 * kDpBuckets * k, *cursor = counts + kDpBuckets; // (cleared with the pass's counters)
                k_dp_sort_count<<<1024, 256, 0, st[k]>>>(sinks.s[1 + k], row_shift, counts, lane_min[k]);
                k_dp_sort_scan<<<1, 256, 0, st[k]>>>(counts, cursor);
                k_dp_sort_place<<<1024, 256, 0, st[k]>>>(sinks.s[1 + k], row_shift, cursor, R.d_dp_order[k], lane_min[k]);
                ord[k] = R.d_dp_order[k];
            }
        }
        if (b1) launch_dp_lane<16>(nw, x2, b1, st[0], cx, sinks.s[1], ord[0], rb, sel, (uint32_t *)R.d_dp_scratch[0], w1, unsup, lane_min[0]);
        k_dp_group<1><<<R.dp_blocks[0], 64, 0, st[0]>>>(cx, sinks.s[1], rb, sel, R.d_dp_scratch[0], R.dp_stride[0], lane_min[0]);
        if (b2) launch_dp_lane<16>(nw, x2, b2, st[1], cx, sinks.s[2], ord[1], rb, sel, (uint32_t *)R.d_dp_scratch[1], w2, unsup, lane_min[1]);
        k_dp_group<4><<<R.dp_blocks[1], 64, 0, st[1]>>>(cx, sinks.s[2], rb, sel, R.d_dp_scratch[1], R.dp_stride[1], lane_min[1]);
        uint32_t *p = R.d_dp_lane;
        launch_dp_lane<8>(nw, x2, R.dp_lane_blocks, side0, cx, sinks.s[4], nullptr, rb, sel, p, lane_short_words(0), unsup, 0u);
        p += lane_short_words(0) * R.dp_lane_blocks;
        launch_dp_lane<16>(nw, x2, R.dp_lane_blocks, side0, cx, sinks.s[0], nullptr, rb, sel, p, lane_short_words(1), unsup, 0u);
        p += lane_short_words(1) * R.dp_lane_blocks;
        launch_dp_lane<16>(nw, x2, R.dp_lane_blocks, side0, cx, sinks.s[5], nullptr, rb, sel, p, lane_short_words(2), unsup, 0u);
        k_dp_sel<16><<<R.dp_blocks[2], 64, 0, side1>>>(cx, sinks.s[3], rb, sel, R.d_dp_scratch[2], R.dp_stride[2]);
    } else {
    k_dp_group<1><<<R.dp_blocks[0], 64, 0, s>>>(cx, sinks.s[1], rb, sel, R.d_dp_scratch[0], R.dp_stride[0], 0xFFFFFFFFu);
    k_dp_small<<<2560, 256, 0, side0>>>(cx, sinks.s[0], rb, sel);
    k_dp_group<4><<<R.dp_blocks[1], 64, 0, side1>>>(cx, sinks.s[2], rb, sel, R.d_dp_scratch[1], R.dp_stride[1], 0xFFFFFFFFu);
    // (the half-wave class behind the 65-256-column class looks like the long pole on a timeline; moved behind the shorter chains
    //  the stage takes the same 3.3-3.4 ms: the kernels share the chip, the stage is the sum of their work)
    k_dp_tiny<<<2048, 256, 0, s>>>(cx, sinks.s[4], rb, sel);
    k_dp_half<<<2048, 256, 0, side1>>>(cx, sinks.s[5], rb, sel);
    k_dp_sel<16><<<R.dp_blocks[2], 64, 0, side0>>>(cx, sinks.s[3], rb, sel, R.d_dp_scratch[2], R.dp_stride[2]);
    }
    for (int k = 0; k < n_side; k++) { HIP_TRY(hipEventRecord(R.dp_join[k], R.dp_stream[k])); HIP_TRY(hipStreamWaitEvent(s, R.dp_join[k], 0)); }
    return 0;
}

static int tier1_error(mcx_ctx *c, const PassRes &R);
static int tier1_pass_end(mcx_ctx *c, const PassRes &T, uint32_t m, mcx_stats *t1, mcx_stats *stats, int e);

// One tier over a selection of pairs.  early (tier 0 only): the pairs that run over the tier's capacities while clustering
// are listed on the device; once the rest of the pass is queued, the large tier maps them on its own stream — its kernels
// are bound by their slowest pair, not by the chip, so they hide behind the pass instead of following it.
// state_off: the pass's pair records start at record state_off of the tier.  queue_only: the kernels are queued on R's stream
// and that is all — the caller goes on and calls pass_finish() for the pass later (*queued = its timing events).
static int pass_finish(mcx_ctx *c, int tier, const PassRes &R, uint32_t n_sel, mcx_stats *stats, bool timing, int e);
static int queue_batch_tail(mcx_ctx *c);
static int tail_reserve(mcx_ctx *c);

static int run_pairs(mcx_ctx *c, int tier, const PassRes &R, const ReadBatch &rb, int paired, PairSel sel, AlnRec *d_recs,
                     uint32_t *d_cig, mcx_stats *stats, bool timing, bool early = false, uint32_t state_off = 0, int *queued = nullptr, bool hits_from_tier0 = false, bool no_n_reads = false)
{
    if (sel.n == 0) return 0;
    hipStream_t s = R.stream;
    const Knobs &kn = c->kn;
    Ctx cx = make_ctx(c, tier, paired);
    cx.state += (size_t)state_off * (size_t)cx.lay.stride;
    cx.seed_pool = R.d_rseeds;
    const int nr = paired ? 2 : 1;
    early = early && tier == 0 && c->overlap_tiers;
    HIP_TRY(hipMemsetAsync(R.d_cnt, 0, CNT_ALL * sizeof(uint32_t), s)); // (the pass's counters, the class counts of its order, the bucket counts of its DP lists)
    SeedOut so; so.tasks = R.d_tasks; so.n_tasks = R.d_cnt + CNT_TASKS; so.task_cap = R.task_cap;
    so.read_ext = c->d_read_ext; so.read_blocks = c->d_read_blocks;
    so.packed = c->packed_now ? c->packed_now : c->d_packed; so.wpad = c->wpad; so.queue = R.d_cnt + CNT_QUEUE;
    so.src_state = nullptr; so.src_lay = c->tier[0].lay; so.src_caps = c->tier[0].caps;
    if (hits_from_tier0 && tier == 1 && cx.ix.sa_full && !kn.late_reseed) so.src_state = c->tier[0].state;
    RescueList rl; rl.ids = R.d_rescue; rl.n = R.d_cnt + CNT_RESCUE; rl.cap = R.rescue_cap;
    EarlyList el; el.ids = nullptr; el.est = nullptr; el.n = R.d_cnt + CNT_EARLY; el.cap = 0; el.n_hits = R.d_cnt + CNT_EARLY_HITS;
    if (early) { el.ids = c->t1.d_sel_ids; el.est = c->t1.d_est; el.cap = (uint32_t)c->max_reads; }
    const bool late = early && c->overlap_late;
    EarlyList ll; ll.ids = nullptr; ll.est = nullptr; ll.n = R.d_cnt + CNT_LATE; ll.cap = 0; ll.n_hits = R.d_cnt + CNT_EARLY_HITS;
    if (late) { ll.ids = c->t2.d_sel_ids; ll.est = c->t2.d_est; ll.cap = kLateRoom; }
    JobSinks sinks;
    for (int k = 0; k < kDpClasses; k++) { sinks.s[k].jobs = R.d_jobs[k]; sinks.s[k].count = R.d_cnt + CNT_JOB0 + k * kCntPad; sinks.s[k].cap = R.job_cap[k]; }
    sinks.unsupported = R.d_cnt + CNT_UNSUP;
    const unsigned pb = (sel.n + 255) / 256;
    int e = 0, rc2 = 0;
    if (timing) HIP_TRY(hipEventRecord(R.ev[e++], s));
    {
        // LDS for the packed reads: words per lane for the longest read x lanes; narrower blocks for long reads
        const int pkw = packed_words(c->rlen_max);
        const int threads = pkw * 256 * 4 <= 48 * 1024 ? 256 : (pkw * 128 * 4 <= 48 * 1024 ? 128 : 64);
        const int rpl = seed_reads_per_lane((uint64_t)sel.n * nr);
        const unsigned blocks_s = std::min<unsigned>((sel.n * nr + threads * rpl - 1) / (threads * rpl), 4096u); // (the queue feeds whatever grid runs)
        k_seed<<<blocks_s, threads, (size_t)pkw * threads * 4, s>>>(cx, rb, sel, so, pkw, rpl, kn.seed_fm_budget);
#ifdef MCX_SEED_STATS
        if (tier == 0 && sel.n > 100000) {
            unsigned long long h[2][24];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_seed_hist), sizeof h);
            fprintf(stderr, "[k_seed] index blocks per read, by bit length of the count: reads");
            for (int k = 0; k < 12; k++) fprintf(stderr, " %llu", h[0][k]);
            fprintf(stderr, " | blocks");
            for (int k = 0; k < 12; k++) fprintf(stderr, " %llu", h[1][k]);
            fprintf(stderr, "\n");
            memset(h, 0, sizeof h);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_seed_hist), h, sizeof h);
        }
#endif
    }
    if (timing) HIP_TRY(hipEventRecord(R.ev[e++], s));
    if (R.d_tasks) k_sa<<<4096, 256, 0, s>>>(cx, so, paired, R.d_cnt + CNT_LF);
    if (timing) HIP_TRY(hipEventRecord(R.ev[e++], s));
    // the pairs in the order of their weight (k_order_*): worth two small passes when the pass is a large one
    const uint32_t *order = nullptr, *order_cnt = nullptr;
    bool split = false; // (time stamps 10 and 11 were taken: the straight-line path's and the order's share of the stage before clustering)
    if (tier == 0 && sel.n >= kn.order_min && c->d_order && !kn.no_work_order) {
        // ahead of them, on a whole batch: the straight-line pairs from their seeds to their records (k_simple); what is left is listed
        // by weight for the per-pair kernels.  (Not without the suffix array in HBM — the seeds must be text positions —, not on a
        // selection: k_simple takes pair = record.  With the -vcf bookkeeping on it writes the pairs' detail records as well.)
        const bool no_simple = kn.no_simple;
        const uint8_t *done = nullptr;
        if (!no_simple && !sel.ids && cx.ix.sa_full && cx.packed) {
            // collect (every pair) -> solve (the problems written down) -> replay (the pairs that wrote some down); MCX_SIMPLE_NO_DP: a pair
            // with such a problem takes the general path
            SimpleLater sl; sl.pairs = kn.simple_no_dp ? nullptr : c->d_sl_pairs; sl.jobs = c->d_sl_jobs; sl.res = c->d_sl_res; sl.job_list = c->d_sl_list;
            sl.n_pairs = R.d_cnt + CNT_SIMPLE_LATER; sl.n_jobs = R.d_cnt + CNT_SIMPLE_JOBS; sl.cap = std::min<uint32_t>(c->sl_cap, std::max<uint32_t>(sel.n / 2, 1024u));
            const unsigned lb = (sl.cap + 255) / 256, jb = std::min<unsigned>((sl.cap * kSimpleJobs + 255) / 256, 3072u); // (launched for what the lists may hold: their lengths stay on the device)
            auto launch = [&](auto nw, auto detail) {
                constexpr bool NW = decltype(nw)::value, DT = decltype(detail)::value;
                k_simple<NW, kDpCollect, DT><<<pb, 256, 0, s>>>(cx, rb, sel.n, sel.est, so.read_blocks, d_recs, c->d_pout, c->d_done, c->d_batch_flags + 2, R.d_cnt + CNT_SIMPLE, sl);
                if (sl.pairs) {
                    k_simple_dp<NW><<<jb, 256, 0, s>>>(cx, sl);
                    k_simple<NW, kDpReplay, DT><<<lb, 256, 0, s>>>(cx, rb, sel.n, sel.est, so.read_blocks, d_recs, c->d_pout, c->d_done, c->d_batch_flags + 2, R.d_cnt + CNT_SIMPLE, sl);
                }
            };
            if (cx.pm.use_nw) { if (cx.detail) launch(std::true_type(), std::true_type()); else launch(std::true_type(), std::false_type()); }
            else { if (cx.detail) launch(std::false_type(), std::true_type()); else launch(std::false_type(), std::false_type()); }
            done = c->d_done;
        }
        if (timing) HIP_TRY(hipEventRecord(R.ev[10], s));
        uint32_t *cls_cnt = R.d_cnt + CNT_ORDER; // (cleared with the pass's counters)
        order_cnt = cls_cnt; // (the class counts: how many pairs the order lists — all of them without k_simple — and where a class begins)
        const unsigned ob = (sel.n + 256 * kOrderTile - 1) / (256 * kOrderTile);
        k_order_count<<<ob, 256, 0, s>>>(sel, so.read_blocks, nr, cls_cnt, done);
        k_order_place<<<ob, 256, 0, s>>>(sel, so.read_blocks, nr, cls_cnt, c->d_order, done);
        order = c->d_order;
        if (timing) { HIP_TRY(hipEventRecord(R.ev[11], s)); split = true; }
    }
    const size_t cl_bytes = cluster_lds_bytes(cx.caps.hit_cap, cx.caps.cand_cap);
    if (tier == 1 && cl_bytes <= 60 * 1024 && !kn.cluster_by_lane) { // the large tier's pairs: a wavefront each, in two launches by size
        const int small = std::min(kClusterSmall, cx.caps.hit_cap);
        k_cluster_wave<<<std::min<unsigned>(sel.n, 16384u), 64, cluster_lds_bytes(small, small), s>>>(cx, rb, sel, rl, so.read_blocks, -1, small, small, small);
        if (small < cx.caps.hit_cap)
            k_cluster_wave<<<std::min<unsigned>(sel.n, 8192u), 64, cl_bytes, s>>>(cx, rb, sel, rl, so.read_blocks, small, 1 << 30, cx.caps.hit_cap, cx.caps.cand_cap);
    } else k_cluster<<<pb, 256, 0, s>>>(cx, rb, sel, rl, so.read_blocks, el, order, order_cnt, 0, kWorkClasses - 1);
    if (early && c->h_early) k_publish_early<<<1, 64, 0, s>>>(R.d_cnt + CNT_EARLY, c->d_batch_flags + 3, c->d_early);
    if (early) HIP_TRY(hipEventRecord(c->ev_clustered, s));
    if (timing) HIP_TRY(hipEventRecord(R.ev[e++], s));
    // mate rescue beside the build of the pairs that do not await it (k_build's modes), when the set has a side stream for it
    const bool rescue_aside = paired && R.dp_stream[0] && !kn.rescue_in_line;
    hipStream_t rs = rescue_aside ? R.dp_stream[0] : s;
    if (rescue_aside) { HIP_TRY(hipEventRecord(R.dp_fork, s)); HIP_TRY(hipStreamWaitEvent(rs, R.dp_fork, 0)); }
    if (paired) {
        {
            RescueWork rw; rw.tasks = R.d_rtasks; rw.res = R.d_rres; rw.seeds = R.d_rseeds; rw.plans = R.d_rplans; rw.n_tasks = R.d_cnt + CNT_RTASK; rw.n_plans = R.d_cnt + CNT_RPLAN; rw.n_seeds = R.d_cnt + CNT_RSEED;
            rw.task_cap = R.rtask_cap; rw.seed_cap = R.rseed_cap; rw.ids_n = R.d_rescue_n; rw.n_ids_n = R.d_cnt + CNT_RESCUE_N;
            RescueList rn; rn.ids = R.d_rescue_n; rn.n = R.d_cnt + CNT_RESCUE_N; rn.cap = R.rescue_cap;
            const unsigned gb = std::min<unsigned>(std::max<unsigned>((sel.n + 255) / 256, 1u), 1024u);
            k_rescue_plan<<<gb, 256, 0, rs>>>(cx, rb, sel, rl, rw);
            if (tier == 0) k_rescue_eval<2048><<<4096, 256, 0, rs>>>(cx, rb, sel, rw);
            else k_rescue_eval<4096><<<4096, 256, 0, rs>>>(cx, rb, sel, rw);
            k_rescue_apply<<<gb, 256, 0, rs>>>(cx, rw);
            // the pairs with an N in a read (none in most batches: the launch then ends at once — once it gets onto the chip, which beside
            // the other tier's long-lived wavefronts took the large tier half a millisecond: left out where the batch is known to hold no N)
            if (no_n_reads) {}
            else if (tier == 0) k_rescue<2048><<<256, kRescueThreads, 0, rs>>>(cx, rb, sel, rn, R.d_kscratch);
            else k_rescue<4096><<<256, kRescueThreads, 0, rs>>>(cx, rb, sel, rn, R.d_kscratch);
        }
    }
    // three time stamps: in line they bracket rescue | nothing | build, with the rescue aside build (others) | what is left of the wait for the rescue | build (its pairs)
    // (the large tier's pairs: a wavefront each — k_build_wave; MCX_BUILD_BY_LANE: a lane each there too)
    const size_t bw_bytes = (size_t)6 * cx.caps.cand_cap * sizeof(int32_t);
    const bool build_wave = tier == 1 && bw_bytes <= 48 * 1024 && !kn.build_by_lane;
    const int build_wave_limit = kn.build_wave_limit;
    auto build = [&](int mode) {
        if (build_wave) k_build_wave<<<std::min<unsigned>(sel.n, 8192u), 64, bw_bytes, s>>>(cx, rb, sel, sinks, R.d_cnt + CNT_CELLS, R.d_cnt + CNT_UNSUP, mode, rl, build_wave_limit);
        else k_build<<<pb, 256, 0, s>>>(cx, rb, sel, sinks, R.d_cnt + CNT_CELLS, R.d_cnt + CNT_UNSUP, ll, order, order_cnt, mode, rl);
    };
    if (rescue_aside) {
        HIP_TRY(hipEventRecord(R.dp_join[0], rs));
        build(1);
        if (timing) HIP_TRY(hipEventRecord(R.ev[e++], s));
        HIP_TRY(hipStreamWaitEvent(s, R.dp_join[0], 0));
        if (timing) HIP_TRY(hipEventRecord(R.ev[e++], s));
        build(2);
    } else {
        if (timing) { HIP_TRY(hipEventRecord(R.ev[e++], s)); HIP_TRY(hipEventRecord(R.ev[e++], s)); }
        build(0);
    }
    if (late) HIP_TRY(hipEventRecord(c->ev_built, s));
    if (timing) HIP_TRY(hipEventRecord(R.ev[e++], s));
    if ((rc2 = launch_dp(kn, R, cx, sinks, rb, sel, c->rlen_max))) return rc2;
    if (timing) HIP_TRY(hipEventRecord(R.ev[e++], s));
    k_finish<<<pb, 256, 0, s>>>(cx, rb, sel, d_recs, c->d_pout, R.d_ov, R.d_cnt + CNT_OV, R.ov_cap, c->d_batch_flags + 2, order, order_cnt);
    if (timing) HIP_TRY(hipEventRecord(R.ev[e++], s));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(R.h_cnt, R.d_cnt, CNT_N * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    const int e_flag = e | (rescue_aside ? 0x100 : 0) | (split ? 0x200 : 0); // (for pass_finish: which stage a time stamp closes)
    if (queued) { *queued = e_flag; return 0; }
    hipEvent_t ev_dbg[2] = {nullptr, nullptr}; // (MCX_TIMING: when tier 0 and the large tier beside it were done)
    if (early && kn.timing) {
        for (int k = 0; k < 2; k++) HIP_TRY(hipEventCreate(&ev_dbg[k]));
        HIP_TRY(hipEventRecord(ev_dbg[0], s));
    }
    uint32_t n_late = 0;
    if (early) { // the pairs k_cluster listed: through the large tier now, while the kernels above run
        const PassRes &T = c->t1;
        uint32_t n_early = 0;
        HIP_TRY(hipStreamWaitEvent(T.stream, c->ev_clustered, 0));
        bool no_n;
        if (c->h_early) { // k_publish_early wrote both numbers into page-locked memory
            HIP_TRY(timed_wait(kn.timing, "wait for k_cluster (the early list's length)", __LINE__, [&] { return hipEventSynchronize(c->ev_clustered); }));
            n_early = c->h_early[0]; no_n = c->h_early[1] == 0;
        } else {
            HIP_TRY(hipMemcpyAsync(T.h_cnt, R.d_cnt + CNT_EARLY, sizeof(uint32_t), hipMemcpyDeviceToHost, T.stream));
            HIP_TRY(hipMemcpyAsync(T.h_cnt + 1, c->d_batch_flags + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, T.stream)); // (k_pack_reads is long done)
            HIP_TRY(hipStreamSynchronize(T.stream));
            n_early = T.h_cnt[0]; no_n = T.h_cnt[1] == 0;
        }
        if (n_early > el.cap) { (void)hipStreamSynchronize(s); return fail(MCX_ERR_CAPACITY, "overflow list overflow"); }
        if (stats) stats->tier1_pairs += n_early;
        const bool t1_timing = kn.timing;
        if (n_early + (late ? kLateRoom : 0) > c->tier[1].max_pairs) { if (int r = tier1_grow(c, (uint64_t)n_early + (late ? kLateRoom : 0))) { (void)hipStreamSynchronize(s); return r; } }
        const uint32_t room = c->tier[1].max_pairs - (late ? kLateRoom : 0); // (the last records are the late list's)
        mcx_stats t1;
        uint32_t m = 0;
        int e1 = -1, e2 = -1;
        for (uint32_t lo = 0; lo < n_early && rc2 == 0; lo += room) {
            // (all but the last pass are waited for here; the last one is left under way while the late list is seen to)
            m = std::min<uint32_t>(room, n_early - lo);
            const bool last = lo + room >= n_early;
            PairSel s1; s1.n = m; s1.ids = T.d_sel_ids + lo; s1.est = T.d_est + lo;
            memset(&t1, 0, sizeof t1);
            rc2 = run_pairs(c, 1, T, rb, paired, s1, d_recs, d_cig, t1_timing ? &t1 : stats, t1_timing, false, 0, last ? &e1 : nullptr, false, no_n);
            if (!last && rc2 == 0) rc2 = tier1_pass_end(c, T, m, t1_timing ? &t1 : nullptr, stats, -1);
        }
        if (late && rc2 == 0) {
            // what ran over after clustering (k_build's list): known once tier 0 has built; a handful of pairs, which take the
            // large tier on the third set of resources while tier 0's DP and finish stages run
            const PassRes &U = c->t2;
            HIP_TRY(timed_wait(kn.timing, "wait for k_build (the late list's length)", __LINE__, [&] { return hipEventSynchronize(c->ev_built); }));
            HIP_TRY(hipMemcpyAsync(U.h_cnt, R.d_cnt + CNT_LATE, sizeof(uint32_t), hipMemcpyDeviceToHost, U.stream));
            HIP_TRY(hipStreamSynchronize(U.stream));
            n_late = std::min<uint32_t>(U.h_cnt[0], kLateRoom);
            if (n_late) {
                PairSel s2; s2.n = n_late; s2.ids = U.d_sel_ids; s2.est = U.d_est;
                // (their seed hits are in this pass's records — pair = record when the pass is a whole batch — and complete: what ran over came later)
                rc2 = run_pairs(c, 1, U, rb, paired, s2, d_recs, d_cig, stats, false, false, c->tier[1].max_pairs - kLateRoom, &e2, sel.ids == nullptr && state_off == 0, no_n);
                if (stats) stats->tier1_pairs += n_late;
            }
        }
        // (everything of the batch is queued: what follows its kernels goes behind them now, before the host waits for any of the passes)
        if (rc2 == 0 && c->tail.want && !sel.ids && state_off == 0) rc2 = queue_batch_tail(c);
        if (e1 >= 0) { const int r = tier1_pass_end(c, T, m, t1_timing ? &t1 : nullptr, stats, e1); if (rc2 == 0) rc2 = r; }
        if (e2 >= 0) { const int r = tier1_pass_end(c, c->t2, n_late, nullptr, stats, e2); if (rc2 == 0) rc2 = r; }
    }
    if (ev_dbg[0]) HIP_TRY(hipEventRecord(ev_dbg[1], c->t1.stream));
    if (!early && tier == 0 && rc2 == 0 && c->tail.want && !sel.ids && state_off == 0) rc2 = queue_batch_tail(c);
    HIP_TRY(timed_wait(kn.timing, "wait for the pass's stream", __LINE__, [&] { return hipStreamSynchronize(s); }));
    if (ev_dbg[0]) {
        float a = 0, b = 0;
        HIP_TRY(hipEventSynchronize(ev_dbg[1]));
        HIP_TRY(hipEventElapsedTime(&a, c->ev_clustered, ev_dbg[0])); HIP_TRY(hipEventElapsedTime(&b, c->ev_clustered, ev_dbg[1]));
        fprintf(stderr, "[run_pairs] after clustering: tier 0 done at %.2f ms, the large tier beside it by %.2f ms\n", a, b);
        for (int k = 0; k < 2; k++) (void)hipEventDestroy(ev_dbg[k]);
    }
    if (rc2) return rc2;
    return pass_finish(c, tier, R, sel.n, stats, timing, e_flag);
}

// what follows a pass once its stream has been joined: the work lists' overflow checks, counts and stage times
static int pass_finish(mcx_ctx *c, int tier, const PassRes &R, uint32_t n_sel, mcx_stats *stats, bool timing, int e)
{
    if (tier == 1 && c->kn.tier1_hist) { // experiments: how heavy are the pairs of the large tier?
        std::vector<PairHdr> hd(n_sel);
        HIP_TRY(hipMemcpy2D(hd.data(), sizeof(PairHdr), c->tier[1].state, (size_t)c->tier[1].lay.stride, sizeof(PairHdr), n_sel, hipMemcpyDeviceToHost));
        const int edges[8] = {16, 32, 64, 128, 256, 512, 1024, 1 << 30};
        uint32_t hh[8] = {0}, hc[8] = {0}, hf[8] = {0};
        for (const PairHdr &h : hd) {
            const int nh = std::max(h.n_hits[0], h.n_hits[1]), nc = std::max(h.n_cands[0], h.n_cands[1]);
            for (int b = 0; b < 8; b++) if (nh <= edges[b]) { hh[b]++; break; }
            for (int b = 0; b < 8; b++) if (nc <= edges[b]) { hc[b]++; break; }
            for (int b = 0; b < 8; b++) if (h.n_frags <= edges[b]) { hf[b]++; break; }
        }
        fprintf(stderr, "[tier 1 hist] %u pairs; per-read maximum <=16,32,64,128,256,512,1024,more: hits", n_sel);
        for (int b = 0; b < 8; b++) fprintf(stderr, " %u", hh[b]);
        fprintf(stderr, " | candidates");
        for (int b = 0; b < 8; b++) fprintf(stderr, " %u", hc[b]);
        fprintf(stderr, " | fragments");
        for (int b = 0; b < 8; b++) fprintf(stderr, " %u", hf[b]);
        fprintf(stderr, "\n");
    }
    const uint32_t *n = R.h_cnt;
    if (c->kn.dp_hist && n_sel >= 1024) { // experiments: the sizes of the pass's DP problems (query x target, by bit length), per list
        static const char *names[kDpClasses] = {"small", "wave1", "wave4", "wave16", "tiny", "half"};
        for (int k = 0; k < kDpClasses; k++) {
            const uint32_t m = std::min(n[CNT_JOB0 + k * kCntPad], R.job_cap[k]);
            if (!m) continue;
            std::vector<DpJob> jobs(m);
            HIP_TRY(hipMemcpy(jobs.data(), R.d_jobs[k], (size_t)m * sizeof(DpJob), hipMemcpyDeviceToHost));
            unsigned long long hist[12][12] = {{0}}, cells = 0, diag = 0;
            for (const DpJob &j : jobs) {
                const int a = std::min(11, j.rLen ? 32 - __builtin_clz((unsigned)j.rLen) : 0), b = std::min(11, j.gLen ? 32 - __builtin_clz((unsigned)j.gLen) : 0);
                hist[a][b]++; cells += (unsigned long long)j.rLen * j.gLen; diag += (unsigned long long)(j.rLen + j.gLen - 1);
            }
            fprintf(stderr, "[dp hist] tier %d list %s: %u problems, %llu cells (mean %.1f), mean anti-diagonals %.1f; rows = bit length of the query, columns = of the target\n", tier, names[k], m, cells,
                    (double)cells / m, (double)diag / m);
            for (int a = 0; a < 12; a++) {
                bool any = false;
                for (int b = 0; b < 12; b++) any |= hist[a][b] != 0;
                if (!any) continue;
                fprintf(stderr, "[dp hist]   q<2^%-2d", a);
                for (int b = 0; b < 12; b++) fprintf(stderr, " %9llu", hist[a][b]);
                fprintf(stderr, "\n");
            }
        }
    }
    if (tier == 0 && !c->dp_grown) c->job2_seen = std::max(c->job2_seen, n[CNT_JOB2]); // (batch_close: dp_scratch_grow)
    // a work list that ran over: nothing of this pass is kept, the caller maps the selection in two halves
    if ((R.d_tasks && n[CNT_TASKS] > R.task_cap) || n[CNT_RESCUE] > R.rescue_cap || n[CNT_RTASK] > R.rtask_cap || n[CNT_RSEED] > R.rseed_cap) return kListOverflow;
    for (int k = 0; k < kDpClasses; k++) if (n[CNT_JOB0 + k * kCntPad] > R.job_cap[k]) return kListOverflow;
    if (n[CNT_UNSUP]) return fail(MCX_ERR_UNSUPPORTED, "a gapped fragment exceeds 2048 x 1024 cells per side");
    if (timing && c->kn.timing)
        fprintf(stderr, "[run_pairs] pairs %u: sa tasks %u, rescue pairs %u (windows %u), dp jobs by class (tiny) %u %u (half) %u %u %u %u, cells %llu, overflow pairs %u (+ %u listed while clustering, %u of them for their seed hits), %u straight-line pairs (k_simple)\n", n_sel,
                n[CNT_TASKS], n[CNT_RESCUE], n[CNT_RTASK], n[CNT_JOB4], n[CNT_JOB0], n[CNT_JOB5], n[CNT_JOB1], n[CNT_JOB2], n[CNT_JOB3], *(const unsigned long long *)(n + CNT_CELLS), n[CNT_OV], n[CNT_EARLY], n[CNT_EARLY_HITS], n[CNT_SIMPLE]);
    if (stats) {
        stats->dp_jobs += (int64_t)n[CNT_JOB0] + n[CNT_JOB1] + n[CNT_JOB2] + n[CNT_JOB3] + n[CNT_JOB4] + n[CNT_JOB5];
        stats->dp_cells += (int64_t)(*(const unsigned long long *)(n + CNT_CELLS));
        stats->simple_pairs += n[CNT_SIMPLE];
        if (timing) {
            float ms[10];
            const bool aside = (e & 0x100) != 0;
            const int n_ev = e & 0xFF;
            for (int i = 0; i + 1 < n_ev; i++) HIP_TRY(hipEventElapsedTime(&ms[i], R.ev[i], R.ev[i + 1]));
            stats->ms_seed += ms[0]; stats->ms_sa += ms[1];
            if (e & 0x200) { // ev[2] .. ev[10]: the straight-line path; ev[10] .. ev[11]: the order; ev[11] .. ev[3]: k_cluster
                float a = 0, b = 0, d = 0;
                HIP_TRY(hipEventElapsedTime(&a, R.ev[2], R.ev[10])); HIP_TRY(hipEventElapsedTime(&b, R.ev[10], R.ev[11])); HIP_TRY(hipEventElapsedTime(&d, R.ev[11], R.ev[3]));
                stats->ms_simple += a; stats->ms_order += b; stats->ms_cluster += d;
            } else stats->ms_cluster += ms[2];
            // (run_pairs: in line ms[3] is the rescue, ms[4] nothing, ms[5] the build; with the rescue on a stream of its own ms[3] and ms[5] are the
            //  build's two launches and ms[4] what was left to wait for the rescue)
            if (aside) { stats->ms_build += ms[3] + ms[5]; stats->ms_rescue += ms[4]; }
            else { stats->ms_rescue += ms[3] + ms[4]; stats->ms_build += ms[5]; }
            stats->ms_dp += ms[6]; stats->ms_finish += ms[7];
        }
    }
    return 0;
}

// end of a pass of the large tier that was left under way beside tier 0 (e: its timing events, -1: it was waited for already)
static int tier1_pass_end(mcx_ctx *c, const PassRes &T, uint32_t m, mcx_stats *t1, mcx_stats *stats, int e)
{
    int rc = 0;
    if (e >= 0) {
        HIP_TRY(hipStreamSynchronize(T.stream));
        rc = pass_finish(c, 1, T, m, t1 ? t1 : stats, t1 != nullptr, e);
    }
    if (t1) {
        fprintf(stderr, "[tier 1, beside tier 0] %u pairs: seed %.2f sa %.2f cluster %.2f rescue %.2f build %.2f dp %.2f finish %.2f ms\n", m, t1->ms_seed, t1->ms_sa,
                t1->ms_cluster, t1->ms_rescue, t1->ms_build, t1->ms_dp, t1->ms_finish);
        if (stats) { stats->dp_jobs += t1->dp_jobs; stats->dp_cells += t1->dp_cells; }
    }
    if (rc == kListOverflow) rc = fail(MCX_ERR_CAPACITY, "work list overflow in tier 1");
    if (rc == 0 && T.h_cnt[CNT_OV]) rc = tier1_error(c, T);
    return rc;
}

// a pair that does not even fit the large tier: say which and why
static int tier1_error(mcx_ctx *c, const PassRes &R)
{
    uint32_t first = 0;
    PairOut po; memset(&po, 0, sizeof po);
    if (hipMemcpy(&first, R.d_ov, sizeof first, hipMemcpyDeviceToHost) == hipSuccess)
        (void)hipMemcpy(&po, c->d_pout + first, sizeof po, hipMemcpyDeviceToHost);
    return fail(MCX_ERR_CAPACITY, "a pair exceeded the tier-1 capacities (pair " + std::to_string(first) + ", overflow flags " + std::to_string(po.flags & kOvAny) +
                                  ": 1 hits 2 candidates 4 fragments 8 ops 16 jobs 32 cigar 64 rescue window)");
}

__global__ void k_fill_i32(int32_t *p, int32_t v, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void k_reduce_stats(const uint32_t *a, const uint32_t *b, uint32_t n, unsigned long long *out)
{
    unsigned long long sa = 0, sb = 0, sh = 0; // extension steps, blocks touched, hits (H of SURVEY.md §8d)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { sa += a[i] & 0x7FFFFFFFu; sb += b[i] & 0xFFFFFu; sh += b[i] >> 20; }
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_down(sa, o, 64); sb += __shfl_down(sb, o, 64); sh += __shfl_down(sh, o, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(out, sa); atomicAdd(out + 1, sb); atomicAdd(out + 2, sh); }
}

// ---- avgDist replay on the device (the host only walks the per-chunk sums) ---------------------
// per chunk of 100 pairs: number of proper pairs and their summed distance (ReadMapping.cpp:527-531)
__global__ void k_chunk_sums(const PairOut *po, const uint32_t *off, uint32_t n_pairs, uint32_t chunk, uint32_t *ok_cnt, uint32_t *dist_sum,
                             uint32_t *len_sum, uint32_t *mapped)
{
    const uint32_t n_chunks = (n_pairs + chunk - 1) / chunk;
    uint32_t m = 0;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += gridDim.x * blockDim.x) {
        uint32_t ok = 0, ds = 0, ls = 0;
        const uint32_t p1 = min(n_pairs, (c + 1) * chunk);
        for (uint32_t p = c * chunk; p < p1; p++) {
            const PairOut o = po[p];
            if (o.pair_ok) { ok++; ds += (uint32_t)o.pair_dist; ls += off[2 * p + 2] - off[2 * p]; } // myReadLengthSum, ReadMapping.cpp:529-530
            m += (uint32_t)o.mapped;
        }
        ok_cnt[c] = ok; dist_sum[c] = ds; len_sum[c] = ls;
    }
    for (int o = 32; o > 0; o >>= 1) m += __shfl_down(m, o, 64);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(mapped, m);
}

// pairs whose chunk estimate lies outside their validity interval -> redo list (avg_replay's test)
__global__ void k_check_est(const PairOut *po, uint32_t n_pairs, uint32_t chunk, const int32_t *est_chunk, uint32_t *redo_ids,
                            int32_t *redo_est, uint32_t *n_redo, uint32_t cap)
{
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pairs; p += gridDim.x * blockDim.x) {
        const PairOut o = po[p];
        const int32_t e = est_chunk[p / chunk];
        const bool ok = (o.flags & kRescueUsedEst) ? o.est == e : (e >= o.est_lo && e <= o.est_hi);
        if (!ok) { const uint32_t at = atomicAdd(n_redo, 1u); if (at < cap) { redo_ids[at] = p; redo_est[at] = e; } }
    }
}

// mcx_avg_walk on the device: the estimate every chunk of the batch is checked against.  The walk is no chain: the estimate before chunk k is
// the run's when the batch began (k = 0, or no more than 1000 proper pairs so far — the count only grows, so nothing moved it yet) or the
// rounded mean over everything before k (ReadMapping.cpp:538-539) — prefix sums of the chunks' pairs and distances.  One block: a stretch of
// chunks per thread, the stretches' sums scanned in LDS.
// (first: the batch is the first of its round — of a run on one stream, every batch —: its first chunk takes cur0 as it is; a batch behind
//  others of the round starts from their totals, tp0 / td0 then hold them too)
__global__ void __launch_bounds__(1024) k_avg_walk(const uint32_t *ok, const uint32_t *ds, uint32_t nc, long long cur0, long long tp0, long long td0, int first, int32_t *est_chunk)
{
    __shared__ long long s_tp[1024], s_td[1024];
    const uint32_t per = (nc + 1023u) / 1024u, lo = min(nc, threadIdx.x * per), hi = min(nc, lo + per);
    long long tp = 0, td = 0;
    for (uint32_t k = lo; k < hi; k++) { tp += ok[k]; td += ds[k]; }
    s_tp[threadIdx.x] = tp; s_td[threadIdx.x] = td;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { // inclusive scan over the stretches' sums
        const long long a = (int)threadIdx.x >= o ? s_tp[threadIdx.x - o] : 0, b = (int)threadIdx.x >= o ? s_td[threadIdx.x - o] : 0;
        __syncthreads();
        s_tp[threadIdx.x] += a; s_td[threadIdx.x] += b;
        __syncthreads();
    }
    tp = tp0 + s_tp[threadIdx.x] - tp; td = td0 + s_td[threadIdx.x] - td; // what lies before this thread's stretch
    for (uint32_t k = lo; k < hi; k++) {
        uint32_t cur = (uint32_t)cur0;
        if ((k > 0 || !first) && tp > 1000) cur = (uint32_t)(int)(1. * (double)td / (double)tp + .5);
        est_chunk[k] = (int32_t)(cur * 1.5);
        tp += ok[k]; td += ds[k];
    }
}

// What follows the kernels of a whole-batch pass, queued behind them BEFORE the host waits for the pass (the large tier's passes beside
// it included: their streams' events) instead of in three round trips after it: the seeding statistics, the per-chunk sums, the walk
// and the check of every pair's estimate against its chunk's, everything the host wants of them in one copy to page-locked memory.
// Used when the pass went the plain way (no halved selection, no pair left over for the large tier afterwards: run_selection);
// the caller walks the sums itself too (a few thousand scalars) and takes the device's list only when both walks agree.
static int tail_reserve(mcx_ctx *c)
{
    mcx_ctx::Tail &t = c->tail;
    const uint32_t want = 8 + 4 + 6 + 4 * (uint32_t)((c->max_reads + kReadChunkSize / 2 - 1) / (kReadChunkSize / 2));
    if (t.cap >= want) return 0;
    HIP_TRY(hipMalloc((void **)&t.d, (size_t)want * 4));
    HIP_TRY(hipHostMalloc((void **)&t.h, (size_t)want * 4));
    t.cap = want;
    for (auto &e : t.ev) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return 0;
}

static int queue_batch_tail(mcx_ctx *c)
{
    BatchRun &br = c->run;
    mcx_ctx::Tail &t = c->tail;
    hipStream_t s = c->stream;
    const uint32_t nc = br.n_chunks, n_reads = br.rb.n_reads;
    if (int rc = tail_reserve(c)) return rc;
    if (c->overlap_tiers) { // the passes of the large tier beside this one write records and outcomes of their own pairs
        HIP_TRY(hipEventRecord(t.ev[0], c->t1.stream)); HIP_TRY(hipStreamWaitEvent(s, t.ev[0], 0));
        if (c->overlap_late) { HIP_TRY(hipEventRecord(t.ev[1], c->t2.stream)); HIP_TRY(hipStreamWaitEvent(s, t.ev[1], 0)); }
    }
    // (the sums in the tail's own words, behind the walk's estimates: the per-read statistics stay as the passes left them — a batch whose tail does not
    //  stand in the end, because pairs went on to the large tier or the selection was halved, reduces them again once every pass has run, like a batch
    //  without a tail: round 5 took the tail's snapshot then, made before those passes searched their reads again)
    unsigned long long *d_sum = (unsigned long long *)(t.d + 2); // counters [2..8): three 64-bit sums
    int32_t *d_est = (int32_t *)(t.d + 8);
    uint32_t *d_ok = t.d + 8 + nc, *d_ds = d_ok + nc, *d_ls = d_ds + nc;
    HIP_TRY(hipMemsetAsync(t.d, 0, 8 * sizeof(uint32_t), s));
    k_reduce_stats<<<256, 256, 0, s>>>(c->d_read_ext, c->d_read_blocks, n_reads, d_sum);
    k_chunk_sums<<<(nc + 255) / 256, 256, 0, s>>>(c->d_pout, br.rb.off, br.n_pairs, kReadChunkSize / 2, d_ok, d_ds, d_ls, t.d + 1);
    if (br.paired) {
        k_avg_walk<<<1, 1024, 0, s>>>(d_ok, d_ds, nc, (long long)t.state0[0], (long long)t.state0[1], (long long)t.state0[2], 1, d_est);
        k_check_est<<<2048, 256, 0, s>>>(c->d_pout, br.n_pairs, kReadChunkSize / 2, d_est, c->d_sel_ids, c->d_est, t.d, c->ov_cap);
    }
    HIP_TRY(hipGetLastError());
    uint32_t *h = t.h;
    HIP_TRY(hipMemcpyAsync(h, t.d, 8 * 4, hipMemcpyDeviceToHost, s));                       // n_redo, mapped, statistics
    HIP_TRY(hipMemcpyAsync(h + 8, c->d_batch_flags, 4 * 4, hipMemcpyDeviceToHost, s));      // the CIGAR pool's words, its overflow flag
    HIP_TRY(hipMemcpyAsync(h + 18, d_ok, (size_t)nc * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h + 18 + nc, d_ds, 2 * (size_t)nc * 4, hipMemcpyDeviceToHost, s)); // (distance sums, then length sums)
    if (br.paired) HIP_TRY(hipMemcpyAsync(h + 18 + 3 * nc, d_est, (size_t)nc * 4, hipMemcpyDeviceToHost, s));
    t.queued = t.ran = true;
    br.d_ok = d_ok; br.d_ds = d_ds;
    return 0;
}

static int profile_keys(mcx_ctx *c);
static int profile_queue(mcx_ctx *c);
static int profile_collect(mcx_ctx *c);
static int archive_append(mcx_ctx *c, mcx_ctx::Archive &a, const SparseRec *d_src, uint64_t n, hipStream_t on = nullptr);
static int profile_foreign(mcx_ctx *c, const uint64_t *h_all, uint64_t n_all);
static int sort_reserve(mcx_ctx *c, uint64_t n);
static int profile_accumulate(mcx_ctx *c, const uint64_t *h_all, uint64_t n_all, uint32_t slot_stride, uint32_t own_slot);

extern "C" void mcx_avg_init(int64_t a[4]) { a[0] = 1000; a[1] = 0; a[2] = 0; a[3] = 0; }

// ReadMapping.cpp:462 / :538-539 over consecutive chunks
extern "C" void mcx_avg_walk(int64_t st[3], const uint32_t *pairs, const uint32_t *dist, uint32_t n_chunks, int32_t *est_chunk)
{
    uint32_t cur = (uint32_t)st[0];
    int64_t tp = st[1], td = st[2];
    for (uint32_t k = 0; k < n_chunks; k++) {
        if (est_chunk) est_chunk[k] = (int32_t)(cur * 1.5);
        tp += pairs[k]; td += dist[k];
        if (tp > 1000) cur = (uint32_t)(int)(1. * td / tp + .5);
    }
    st[0] = cur; st[1] = tp; st[2] = td;
}

// runs the tiers for the pairs in `ids` (null: all pairs of the batch) with per-pair estimates
__global__ void k_gather_pout(const PairOut *pout, const uint32_t *ids, uint32_t n, PairOut *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pout[ids[i]];
}

static int run_selection(mcx_ctx *c, const ReadBatch &rb, int paired, const std::vector<uint32_t> *ids,
                         const std::vector<int32_t> *est, int32_t est_all, uint32_t n_pairs, AlnRec *d_recs,
                         uint32_t *d_cig, mcx_stats *stats, bool timing)
{
    hipStream_t s = c->stream;
    const uint32_t n = ids ? (uint32_t)ids->size() : n_pairs;
    if (n == 0) return 0;
    PairSel sel; sel.n = n; sel.ids = nullptr; sel.est = c->d_est;
    if (ids) {
        HIP_TRY(hipMemcpyAsync(c->d_sel_ids, ids->data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        sel.ids = c->d_sel_ids;
    }
    if (est) HIP_TRY(hipMemcpyAsync(c->d_est, est->data(), n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    else k_fill_i32<<<(n + 255) / 256, 256, 0, s>>>(c->d_est, est_all, n);
    const PassRes R0 = res_tier0(c);
    int rc = run_pairs(c, 0, R0, rb, paired, sel, d_recs, d_cig, stats, timing, true);
    // (the batch's tail, if the pass queued it, stands only when the pass is all there was: no halves, no pairs left for the large tier)
    if (rc != 0 || c->h_cnt[CNT_OV] != 0) c->tail.queued = false;
    if (rc == kListOverflow) {
        // unusually many hits or DP problems per read (e.g. indel-heavy long reads): halve the selection
        if (n < 2) return fail(MCX_ERR_CAPACITY, "work list overflow for a single pair");
        if (stats) stats->halved_selections++;
        for (int half = 0; half < 2; half++) {
            const uint32_t lo = half ? n / 2 : 0, hi = half ? n : n / 2;
            std::vector<uint32_t> sub_ids(hi - lo);
            std::vector<int32_t> sub_est(hi - lo);
            for (uint32_t i = lo; i < hi; i++) { sub_ids[i - lo] = ids ? (*ids)[i] : i; sub_est[i - lo] = est ? (*est)[i] : est_all; }
            if ((rc = run_selection(c, rb, paired, &sub_ids, &sub_est, 0, n_pairs, d_recs, d_cig, stats, timing))) return rc;
        }
        return 0;
    }
    if (rc) return rc;
    uint32_t n_ov = c->h_cnt[CNT_OV];
    if (n_ov == 0) return 0;
    if (n_ov > c->ov_cap) return fail(MCX_ERR_CAPACITY, "overflow list overflow");
    // tier 1: the overflowed pairs again, with capacities that are hard bounds for the read length
    std::vector<uint32_t> ov(n_ov);
    HIP_TRY(hipMemcpy(ov.data(), c->d_ov, n_ov * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::sort(ov.begin(), ov.end());
    std::vector<int32_t> ov_est(n_ov);
    // their estimates (and flags): gathered on the device — the whole PairOut array is 128 MB at 4 M pairs
    std::vector<PairOut> ov_out(n_ov);
    for (uint32_t lo = 0; lo < n_ov; lo += kPoutSel) {
        const uint32_t m = std::min<uint32_t>(kPoutSel, n_ov - lo);
        HIP_TRY(hipMemcpyAsync(c->d_sel_ids, ov.data() + lo, m * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        k_gather_pout<<<(m + 255) / 256, 256, 0, s>>>(c->d_pout, c->d_sel_ids, m, c->d_pout_sel);
        HIP_TRY(hipMemcpyAsync(ov_out.data() + lo, c->d_pout_sel, m * sizeof(PairOut), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    for (uint32_t i = 0; i < n_ov; i++) ov_est[i] = ov_out[i].est;
    if (stats) stats->tier1_pairs += n_ov;
    if (c->kn.timing) { // what sent them here
        uint32_t by_flag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t i = 0; i < n_ov; i++) for (int b = 0; b < 8; b++) if (ov_out[i].flags & (1u << b)) by_flag[b]++;
        fprintf(stderr, "[tier 1] %u pairs over the tier-0 capacities: hits %u candidates %u fragments %u ops %u jobs %u cigar %u rescue window %u detail %u\n", n_ov, by_flag[0],
                by_flag[1], by_flag[2], by_flag[3], by_flag[4], by_flag[5], by_flag[6], by_flag[7]);
    }
    // (the pairs that ran over while clustering went through the large tier beside this pass already — run_pairs; what is
    //  left ran over later: column strings, job lists)
    for (uint32_t lo = 0; lo < n_ov; lo += c->tier[1].max_pairs) {
        uint32_t m = std::min<uint32_t>(c->tier[1].max_pairs, n_ov - lo);
        HIP_TRY(hipMemcpyAsync(c->d_sel_ids, ov.data() + lo, m * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->d_est, ov_est.data() + lo, m * sizeof(int32_t), hipMemcpyHostToDevice, s));
        PairSel s1; s1.n = m; s1.ids = c->d_sel_ids; s1.est = c->d_est;
        if (c->kn.timing) { // where the time of the large-capacity tier goes (not added to the caller's stage times)
            mcx_stats t1; memset(&t1, 0, sizeof t1);
            rc = run_pairs(c, 1, R0, rb, paired, s1, d_recs, d_cig, &t1, true);
            fprintf(stderr, "[tier 1] %u pairs: seed %.2f sa %.2f cluster %.2f rescue %.2f build %.2f dp %.2f finish %.2f ms\n", m, t1.ms_seed, t1.ms_sa, t1.ms_cluster,
                    t1.ms_rescue, t1.ms_build, t1.ms_dp, t1.ms_finish);
            if (stats) { stats->dp_jobs += t1.dp_jobs; stats->dp_cells += t1.dp_cells; }
        } else
        rc = run_pairs(c, 1, R0, rb, paired, s1, d_recs, d_cig, stats, false);
        if (rc == kListOverflow) return fail(MCX_ERR_CAPACITY, "work list overflow in tier 1");
        if (rc) return rc;
        if (c->h_cnt[CNT_OV]) return tier1_error(c, R0);
    }
    return 0;
}

// longest read of the batch (a read past max_read_len would overrun the per-read slots of every stage)
__global__ void k_max_read_len(const uint32_t *off, uint32_t n_reads, uint32_t *out)
{
    uint32_t m = 0;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) m = max(m, off[r + 1] - off[r]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

__global__ void k_zero_but(uint32_t *cnt, int n, int keep_a, int keep_b)
{
    for (int k = threadIdx.x; k < n; k += blockDim.x) if (k != keep_a && k != keep_b) cnt[k] = 0;
}

// a batch's reads into the context's own copy (bases 16 bytes at a time: d_bases is 16-byte aligned, the copy's room ends on a multiple of 16)
__global__ void k_keep_reads(const uint8_t *__restrict__ bases, const uint32_t *__restrict__ off, uint32_t n_reads, uint8_t *__restrict__ to_bases, uint32_t *__restrict__ to_off)
{
    const uint64_t total = off[n_reads], n16 = (total + 15) / 16, T = (uint64_t)gridDim.x * blockDim.x, t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = t; i < n16; i += T) ((U4 *)to_bases)[i] = ((const U4 *)bases)[i];
    for (uint64_t i = t; i <= n_reads; i += T) to_off[i] = off[i];
}

// tier 0's pair records: one per pair of the largest batch of this kind (allocated by the first batch that needs them, or ahead of it by mcx_ctx_create_fit)
static int reserve_tier0(mcx_ctx *c, int paired, uint64_t n_reads)
{
    const uint64_t half = (c->max_reads + 1) / 2;
    const uint64_t want = (paired || n_reads <= half) ? half : c->max_reads; // (the single-end tail of an interleaved file fits the pairs' records)
    if (c->tier[0].max_pairs < want) {
        HIP_TRY(hipDeviceSynchronize());
        if (c->tier[0].state) { (void)hipFree(c->tier[0].state); c->tier[0].state = nullptr; c->tier[0].max_pairs = 0; }
        if (int rc = dmalloc(&c->tier[0].state, (size_t)c->tier[0].lay.stride * want)) return rc;
        c->tier[0].max_pairs = (uint32_t)want;
    }
    return 0;
}

extern "C" int mcx_batch_begin(mcx_ctx *c, const uint8_t *d_bases, const uint32_t *d_off, uint32_t n_reads, int paired, int32_t est0,
                               int64_t read_base, mcx_aln *d_aln, uint32_t *d_cigar, mcx_stats *stats)
{
    static_assert(sizeof(mcx_aln) == sizeof(AlnRec), "mcx_aln and AlnRec must have one layout");
    if (!c) return fail(MCX_ERR_ARG, "mcx_batch_begin: null argument");
    // what mcx_stream_next said of the batch it gave out: taken here, before anything can return, so that no later batch inherits it; it counts for the slot's own buffers only
    const bool vouched = c->lens_checked && d_off && d_off == c->lens_checked_off && d_bases == c->lens_checked_bases;
    c->lens_checked = false; c->lens_checked_off = nullptr; c->lens_checked_bases = nullptr;
    const mcx_ctx::PrePacked pre = c->pre;
    c->pre = mcx_ctx::PrePacked();
    if (!d_bases || !d_off || !d_aln || !d_cigar) return fail(MCX_ERR_ARG, "mcx_batch_begin: null argument");
    BatchRun &br = c->run;
    br.open = false;
    if (n_reads == 0) return fail(MCX_ERR_ARG, "mcx_batch_begin: empty batch");
    if (n_reads > c->max_reads) return fail(MCX_ERR_ARG, "batch larger than max_batch_reads");
    if (paired && (n_reads & 1)) return fail(MCX_ERR_ARG, "paired batch with an odd number of reads");
    if (paired && (read_base & 1)) return fail(MCX_ERR_ARG, "paired batch at an odd read_base");
    if ((uintptr_t)d_bases & 15) return fail(MCX_ERR_ARG, "mcx_batch_begin: d_bases must be 16-byte aligned");
    // (refused before anything is mapped or counted: readCount would move in profile_keys / profile_foreign before profile_accumulate refuses)
    if (c->prof_planes && c->prof_settled) return fail(MCX_ERR_ARG, "the profile has been settled (mcx_profile_settle / _finalize): attach it again before mapping more reads");
    HIP_TRY(hipSetDevice(c->idx->device));
    br.t0 = std::chrono::steady_clock::now();
    hipStream_t s = c->stream;
    if (int rc = reserve_tier0(c, paired, n_reads)) return rc;
    br.rb.bases = d_bases; br.rb.off = d_off; br.rb.n_reads = n_reads;
    br.paired = paired; br.read_base = read_base;
    br.recs = (AlnRec *)d_aln; br.cig = d_cigar; br.stats = stats;
    br.cig_cap = (uint32_t)std::min<uint64_t>((uint64_t)MCX_CIGAR_POOL_WORDS(n_reads), 0xFFFFFFFFu); br.cig_words = 0;
    br.n_pairs = paired ? n_reads / 2 : n_reads;
    br.n_chunks = (br.n_pairs + kReadChunkSize / 2 - 1) / (kReadChunkSize / 2);
    br.mapped = 0; br.sums_valid = false; br.d_ok = br.d_ds = nullptr;
    c->tail.queued = c->tail.ran = false;
    HIP_TRY(hipMemsetAsync(c->d_batch_flags, 0, 4 * sizeof(uint32_t), s));
    c->last_paired = paired ? 1 : 0;
    if (vouched) c->h_cnt[1] = (uint32_t)c->rlen_max; // (vouched for: no kernel, no wait at the start of the step — under the copies of the neighbouring batches such a wait takes milliseconds)
    else { // every read must fit the slots the context was sized for
        k_max_read_len<<<512, 256, 0, s>>>(d_off, n_reads, c->d_batch_flags + 1);
        HIP_TRY(hipMemcpyAsync(c->h_cnt, c->d_batch_flags, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (c->h_cnt[1] > (uint32_t)c->rlen_max)
            return fail(MCX_ERR_UNSUPPORTED, "a read of " + std::to_string(c->h_cnt[1]) + " bases is longer than max_read_len (" + std::to_string(c->rlen_max) + ")");
    }
    { // the batch in 2-bit form, once; every seeding pass (tiers, replay) reads it
        HIP_TRY(hipEventRecord(c->ev_pack[0], s));
        const int tpr = ((int)c->h_cnt[1] + 31) / 32 + 1; // (threads per read: for the batch's longest read, found above)
        br.longest = c->h_cnt[1];
        const uint64_t threads = (uint64_t)n_reads * (uint64_t)tpr;
        // (the reads' flag bytes of the -vcf bookkeeping start here: bit 1 — a byte that is not an upper-case ACGT — is k_pack_reads'; MCX_PROF_BY_COLUMN:
        //  tests — every read is treated as if it held one, so that exact seeds are walked column by column like every other fragment)
        if (c->prof_planes) HIP_TRY(hipMemsetAsync(c->d_admit, c->kn.prof_by_column ? 2 : 0, ((size_t)n_reads + 3) & ~(size_t)3, s));
        if (vouched && pre.packed && pre.bases == d_bases && pre.paired == (paired ? 1 : 0) && !c->prof_planes) {
            // packed on its way in, under the batch before it (mcx_stream_submit_packed): the step starts with the search; the any-N word comes along
            c->packed_now = pre.packed;
            HIP_TRY(hipMemcpyAsync(c->d_batch_flags + 3, pre.any_n, sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        } else {
            c->packed_now = c->d_packed;
            k_pack_reads<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(br.rb, paired, c->wpad, tpr, c->d_packed, c->d_batch_flags + 3, c->prof_planes ? c->d_admit : nullptr);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev_pack[1], s));
    }
    c->later.kept_now = false;
    if (c->prof_planes && c->later.have && c->tail.want) { // (mcx_map_batch_dev's batches: one shard.)  The batch's reads for its bookkeeping, which runs when this call has long returned: copied beside the first kernels
        auto &L = c->later;
        L.kept_now = true;
        HIP_TRY(hipEventRecord(L.begun, s)); // (behind what the stream waits for: the batch's reads are in place)
        HIP_TRY(hipStreamWaitEvent(L.keep_stream, L.begun, 0));
        k_keep_reads<<<2048, 256, 0, L.keep_stream>>>(d_bases, d_off, n_reads, L.d_keep_bases, L.d_keep_off);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(L.kept, L.keep_stream));
    }
    int rc;
    br.ms_setup = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - br.t0).count();
    rc = run_selection(c, br.rb, paired, nullptr, nullptr, est0, br.n_pairs, br.recs, d_cigar, stats, true);
    br.t_begun = std::chrono::steady_clock::now();
    if (rc) return rc;
    if (c->tail.queued) memcpy(c->h_cnt + CNT_N, c->tail.h + 2, sizeof br.hs); // (the pass queued the batch's tail and it stands: the statistics came with it)
    else { // seeding statistics (E, blocks, H of SURVEY.md 8d) before the per-read arrays are reused for the chunk sums
        unsigned long long *d_sum = (unsigned long long *)c->d_cnt;
        HIP_TRY(hipMemsetAsync(c->d_cnt, 0, CNT_N * sizeof(uint32_t), s));
        k_reduce_stats<<<256, 256, 0, s>>>(c->d_read_ext, c->d_read_blocks, n_reads, d_sum);
        HIP_TRY(hipMemcpyAsync(c->h_cnt + CNT_N, d_sum, sizeof br.hs, hipMemcpyDeviceToHost, s)); // (read by batch_close, behind the sums' synchronisation: no round trip of its own)
    }
    br.open = true;
    return 0;
}

// Replay of the reference's avgDist feedback (ReadMapping.cpp:462, :538-539).  The
// device reduces the batch to per-chunk sums; the caller walks the chunk trajectory (a few thousand
// scalars; over all shards when there are several); the device lists the pairs whose chunk estimate
// falls outside their validity interval; those are re-run with the exact estimate until none is left.
extern "C" int mcx_batch_sums(mcx_ctx *c, uint32_t *n_chunks, const uint32_t **pairs, const uint32_t **dist, const uint32_t **len)
{
    if (!c || !c->run.open) return fail(MCX_ERR_ARG, "mcx_batch_sums: no batch in flight");
    BatchRun &br = c->run;
    hipStream_t s = c->stream;
    HIP_TRY(hipSetDevice(c->idx->device));
    const uint32_t nc = br.n_chunks;
    if (br.sums_valid && br.ok.size() == nc && !c->kn.no_sums_cache) { // nothing was re-run since the last call (the closing round of a sharded step): the sums still stand
        if (n_chunks) *n_chunks = nc;
        if (pairs) *pairs = br.ok.data();
        if (dist) *dist = br.ds.data();
        if (len) *len = br.ds.data() + nc;
        return 0;
    }
    uint32_t *d_ok = c->d_read_ext, *d_ds = c->d_read_blocks; // the per-read stat arrays were reduced by mcx_batch_begin
    uint32_t *d_ls = d_ds + nc;                               // (2 * n_chunks <= n_reads)
    br.ok.resize(nc); br.ds.resize(2 * (size_t)nc);
    HIP_TRY(hipMemsetAsync(c->d_cnt, 0, CNT_N * sizeof(uint32_t), s));
    k_chunk_sums<<<(nc + 255) / 256, 256, 0, s>>>(c->d_pout, br.rb.off, br.n_pairs, kReadChunkSize / 2, d_ok, d_ds, d_ls, c->d_cnt + CNT_LF);
    HIP_TRY(hipMemcpyAsync(br.ok.data(), d_ok, nc * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(br.ds.data(), d_ds, 2 * (size_t)nc * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(c->h_cnt, c->d_cnt, CNT_N * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    br.mapped = c->h_cnt[CNT_LF];
    br.sums_valid = true; br.d_ok = d_ok; br.d_ds = d_ds;
    if (n_chunks) *n_chunks = nc;
    if (pairs) *pairs = br.ok.data();
    if (dist) *dist = br.ds.data();
    if (len) *len = br.ds.data() + nc;
    return 0;
}

static int replay_listed(mcx_ctx *c, uint32_t n_redo, mcx_stats *stats);
extern "C" int mcx_batch_replay(mcx_ctx *c, const int32_t *est_chunk, uint32_t *n_redone, mcx_stats *stats)
{
    if (!c || !c->run.open || !est_chunk) return fail(MCX_ERR_ARG, "mcx_batch_replay: no batch in flight");
    BatchRun &br = c->run;
    hipStream_t s = c->stream;
    HIP_TRY(hipSetDevice(c->idx->device));
    if (n_redone) *n_redone = 0;
    if (!br.paired) return 0;
    int32_t *d_est_chunk = (int32_t *)c->d_read_ext; // the sums are on the host
    HIP_TRY(hipMemcpyAsync(d_est_chunk, est_chunk, br.n_chunks * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(c->d_cnt, 0, CNT_N * sizeof(uint32_t), s));
    k_check_est<<<2048, 256, 0, s>>>(c->d_pout, br.n_pairs, kReadChunkSize / 2, d_est_chunk, c->d_sel_ids, c->d_est, c->d_cnt + CNT_OV, c->ov_cap);
    HIP_TRY(hipMemcpyAsync(c->h_cnt, c->d_cnt, CNT_N * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint32_t n_redo = c->h_cnt[CNT_OV];
    if (n_redo == 0) return 0;
    if (int rc = replay_listed(c, n_redo, stats)) return rc;
    if (n_redone) *n_redone = n_redo;
    return 0;
}

// the pairs k_check_est listed (d_sel_ids, d_est: n_redo of them) through the path again, with their chunk's estimate
static int replay_listed(mcx_ctx *c, uint32_t n_redo, mcx_stats *stats)
{
    BatchRun &br = c->run;
    br.sums_valid = false;
    c->tail.queued = false; // (what the batch's tail brought is no longer the batch's state)
    if (n_redo > c->ov_cap) return fail(MCX_ERR_CAPACITY, "avgDist replay: redo list overflow");
    if (stats) stats->replayed_pairs += (int64_t)n_redo;
    std::vector<uint32_t> redo(n_redo); std::vector<int32_t> redo_est(n_redo);
    HIP_TRY(hipMemcpy(redo.data(), c->d_sel_ids, n_redo * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(redo_est.data(), c->d_est, n_redo * 4, hipMemcpyDeviceToHost));
    // the list was appended with atomics: bring it into pair order (results do not depend on it)
    std::vector<std::pair<uint32_t, int32_t>> ord(n_redo);
    for (uint32_t i = 0; i < n_redo; i++) ord[i] = std::make_pair(redo[i], redo_est[i]);
    std::sort(ord.begin(), ord.end());
    for (uint32_t i = 0; i < n_redo; i++) { redo[i] = ord[i].first; redo_est[i] = ord[i].second; }
    return run_selection(c, br.rb, br.paired, &redo, &redo_est, 0, br.n_pairs, br.recs, br.cig, stats, false);
}

// The same feedback for a run on several shards with nothing but totals on the wire: a shard's chunks are checked against the
// trajectory that starts from the round's state plus the totals of the shards before it — the walk has a closed form (k_avg_walk), so nobody
// needs anybody else's chunks.
extern "C" int mcx_batch_totals(mcx_ctx *c, int64_t totals[2])
{
    if (!c || !totals) return fail(MCX_ERR_ARG, "mcx_batch_totals: null argument");
    totals[0] = totals[1] = 0;
    if (!c->run.open) return fail(MCX_ERR_ARG, "mcx_batch_totals: no batch in flight");
    uint32_t nc = 0;
    const uint32_t *ok = nullptr, *ds = nullptr;
    if (int rc = mcx_batch_sums(c, &nc, &ok, &ds, nullptr)) return rc;
    for (uint32_t k = 0; k < nc; k++) { totals[0] += ok[k]; totals[1] += ds[k]; }
    return 0;
}

extern "C" int mcx_batch_check(mcx_ctx *c, const int64_t state_before[3], int first_of_round, uint32_t *n_redone, mcx_stats *stats)
{
    if (!c || !state_before || !c->run.open) return fail(MCX_ERR_ARG, "mcx_batch_check: no batch in flight");
    BatchRun &br = c->run;
    if (n_redone) *n_redone = 0;
    if (!br.paired) return 0;
    HIP_TRY(hipSetDevice(c->idx->device));
    int rc;
    if ((!br.sums_valid || !br.d_ok) && (rc = mcx_batch_sums(c, nullptr, nullptr, nullptr, nullptr))) return rc; // (the chunk sums lie at br.d_ok / br.d_ds behind it)
    if ((rc = tail_reserve(c))) return rc;
    hipStream_t s = c->stream;
    int32_t *d_est = (int32_t *)(c->tail.d + 8);
    HIP_TRY(hipMemsetAsync(c->tail.d, 0, 8 * sizeof(uint32_t), s));
    k_avg_walk<<<1, 1024, 0, s>>>(br.d_ok, br.d_ds, br.n_chunks, (long long)state_before[0], (long long)state_before[1], (long long)state_before[2], first_of_round ? 1 : 0, d_est);
    k_check_est<<<2048, 256, 0, s>>>(c->d_pout, br.n_pairs, kReadChunkSize / 2, d_est, c->d_sel_ids, c->d_est, c->tail.d, c->ov_cap);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->tail.h, c->tail.d, 8 * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint32_t n_redo = c->tail.h[0];
    if (n_redo == 0) return 0;
    if ((rc = replay_listed(c, n_redo, stats))) return rc;
    if (n_redone) *n_redone = n_redo;
    return 0;
}

// the state after a round whose batches held `pairs` proper pairs at the summed distance `dist` (mcx_avg_walk's end state in closed form)
extern "C" void mcx_avg_advance(int64_t st[3], int64_t pairs, int64_t dist, int64_t n_chunks)
{
    st[1] += pairs; st[2] += dist;
    if (n_chunks > 0 && st[1] > 1000) st[0] = (int64_t)(uint32_t)(int)(1. * (double)st[2] / (double)st[1] + .5);
}

// what both ways of closing a batch share: the long-CIGAR pool, statistics
static int batch_close(mcx_ctx *c, mcx_stats *stats)
{
    BatchRun &br = c->run;
    int rc = 0;
    if (!br.sums_valid && (rc = mcx_batch_sums(c, nullptr, nullptr, nullptr, nullptr))) return rc;
    {
        uint32_t fl[4] = {0, 0, 0, 0};
        if (c->tail.queued) memcpy(fl, c->tail.h + 8, sizeof fl); // (nothing ran since the batch's tail was copied back)
        else HIP_TRY(hipMemcpy(fl, c->d_batch_flags, sizeof fl, hipMemcpyDeviceToHost));
        if (fl[2] || fl[0] > br.cig_cap) return fail(MCX_ERR_CAPACITY, "the batch's CIGAR pool (" + std::to_string(MCX_CIGAR_STRIDE) + " operations per read on average + " + std::to_string(MCX_CIGAR_SLACK) + ") ran over");
        br.cig_words = fl[0];
    }
    if (stats) {
        int64_t pairs = 0, dist_sum = 0, len_sum = 0;
        if (br.paired) for (uint32_t k = 0; k < br.n_chunks; k++) { pairs += br.ok[k]; dist_sum += br.ds[k]; len_sum += br.ds[br.n_chunks + k]; }
        stats->reads += br.rb.n_reads; stats->mapped += br.mapped; stats->pairs += pairs; stats->pair_dist_sum += dist_sum; stats->pair_len_sum += len_sum;
        memcpy(br.hs, c->h_cnt + CNT_N, sizeof br.hs);
        stats->fm_ext_steps += (int64_t)br.hs[0]; stats->fm_blocks += (int64_t)br.hs[1]; stats->sa_hits += (int64_t)br.hs[2];
        float ms_pack = 0;
        if (hipEventElapsedTime(&ms_pack, c->ev_pack[0], c->ev_pack[1]) == hipSuccess) stats->ms_encode += ms_pack; // k_pack_reads
    }
    if (!c->dp_grown && c->job2_seen) return dp_scratch_grow(c, c->job2_seen); // (the batch is through: nothing holds the scratch)
    return 0;
}

extern "C" int mcx_batch_end(mcx_ctx *c, mcx_stats *stats)
{
    if (!c || !c->run.open) return fail(MCX_ERR_ARG, "mcx_batch_end: no batch in flight");
    HIP_TRY(hipSetDevice(c->idx->device));
    int rc = batch_close(c, stats);
    if (rc == 0 && c->prof_planes && c->later.have && c->later.kept_now) rc = profile_queue(c); // one shard: the batch's own keys decide the duplicate cap — behind the batch, under the next one's kernels
    else if (rc == 0 && c->prof_planes && (rc = profile_collect(c)) == 0) {
        const auto t0 = std::chrono::steady_clock::now();
        rc = profile_keys(c);
        const auto t1 = std::chrono::steady_clock::now();
        if (rc == 0) rc = profile_accumulate(c, nullptr, 0, 0, 0);
        if (c->kn.timing) {
            auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "[mcx profile] mapping %.2f ms, keys %.2f ms, accumulate %.2f ms\n", ms(c->run.t0, t0), ms(t0, t1), ms(t1, std::chrono::steady_clock::now()));
        }
    }
    c->run.open = false;
    if (rc == 0 && stats) stats->ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c->run.t0).count();
    return rc;
}

extern "C" int mcx_batch_end_keys(mcx_ctx *c, mcx_stats *stats, const uint64_t **keys, uint64_t *n_keys)
{
    if (!c || !c->run.open || !keys || !n_keys) return fail(MCX_ERR_ARG, "mcx_batch_end_keys: no batch in flight");
    if (!c->prof_planes) return fail(MCX_ERR_ARG, "mcx_batch_end_keys: no profile attached");
    HIP_TRY(hipSetDevice(c->idx->device));
    if (int e = profile_collect(c)) { c->run.open = false; return e; }
    int rc = batch_close(c, stats);
    if (rc == 0) rc = profile_keys(c);
    if (rc) { c->run.open = false; return rc; }
    // the valid keys sort to the front
    if (c->h_keys_cap < c->run.n_keys) {
        mcx_pinned_free(c->h_keys);
        c->h_keys_cap = std::max<uint64_t>(c->run.n_keys, c->max_reads);
        c->h_keys = (uint64_t *)mcx_pinned_alloc(c->h_keys_cap * sizeof(uint64_t));
        if (!c->h_keys) { c->h_keys_cap = 0; c->run.open = false; return fail(MCX_ERR_DEVICE, "cannot allocate pinned host memory"); }
    }
    if (c->run.n_keys) HIP_TRY(hipMemcpy(c->h_keys, c->run.d_sorted_keys, c->run.n_keys * sizeof(uint64_t), hipMemcpyDeviceToHost));
    *keys = c->h_keys; *n_keys = c->run.n_keys;
    c->run.keys_out = true;
    return 0;
}

extern "C" int mcx_batch_accumulate(mcx_ctx *c, const uint64_t *all_keys, uint64_t n_all, uint32_t slot_stride, uint32_t own_slot)
{
    if (!c || !c->prof_planes) return fail(MCX_ERR_ARG, "mcx_batch_accumulate: no profile attached");
    HIP_TRY(hipSetDevice(c->idx->device));
    if (int e = profile_collect(c)) return e;
    if (own_slot == 0xFFFFFFFFu) { // a shard without reads in this round: the others' admissions still count
        if (c->run.open) return fail(MCX_ERR_ARG, "mcx_batch_accumulate: a batch is in flight");
        return profile_foreign(c, all_keys, n_all);
    }
    if (!c->run.open || !c->run.keys_out) return fail(MCX_ERR_ARG, "mcx_batch_accumulate: call mcx_batch_end_keys first");
    int rc = profile_accumulate(c, all_keys, n_all, slot_stride, own_slot);
    c->run.open = false; c->run.keys_out = false;
    if (rc == 0 && c->run.stats) c->run.stats->ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c->run.t0).count();
    return rc;
}

extern "C" int mcx_map_batch_dev(mcx_ctx *c, const uint8_t *d_bases, const uint32_t *d_off, uint32_t n_reads, int paired,
                                 int64_t avg[4], mcx_aln *d_aln, uint32_t *d_cigar, mcx_stats *stats)
{
    if (!c) return fail(MCX_ERR_ARG, "mcx_map_batch_dev: null argument");
    auto unvouch = [&]() { c->lens_checked = false; c->lens_checked_off = nullptr; c->lens_checked_bases = nullptr; c->pre = mcx_ctx::PrePacked(); }; // (a return before mcx_batch_begin: nothing said of this batch outlives it)
    if (!d_bases || !d_off || !d_aln || !d_cigar || !avg) { unvouch(); return fail(MCX_ERR_ARG, "mcx_map_batch_dev: null argument"); }
    if (n_reads == 0) { unvouch(); return 0; }
    if (paired && (avg[3] % kReadChunkSize)) { unvouch(); return fail(MCX_ERR_ARG, "batches must start on a 200-read chunk boundary"); }
    // (one stream, one trajectory: what follows the batch's kernels — statistics, per-chunk sums, the walk, the check of every pair's estimate —
    //  is queued behind them by the pass itself, queue_batch_tail)
    c->tail.want = true; c->tail.state0[0] = avg[0]; c->tail.state0[1] = avg[1]; c->tail.state0[2] = avg[2];
    const auto t_in = std::chrono::steady_clock::now();
    int rc = mcx_batch_begin(c, d_bases, d_off, n_reads, paired, (int32_t)((uint32_t)avg[0] * 1.5), avg[3], d_aln, d_cigar, stats);
    c->tail.want = false;
    if (rc) return rc;
    int64_t after[3] = {avg[0], avg[1], avg[2]};
    bool tail = c->tail.queued;
    if (tail) { // the sums are here already (and the mapped reads' count, the CIGAR pool's state)
        BatchRun &br = c->run;
        const uint32_t nc = br.n_chunks, *h = c->tail.h;
        br.ok.assign(h + 18, h + 18 + nc); br.ds.assign(h + 18 + nc, h + 18 + 3 * nc);
        br.mapped = h[1]; br.sums_valid = true;
    }
    if (paired) {
        std::vector<int32_t> est(c->run.n_chunks);
        for (int iter = 0;; iter++) {
            uint32_t nc = 0, n_redo = 0;
            const uint32_t *ok = nullptr, *ds = nullptr;
            if ((rc = mcx_batch_sums(c, &nc, &ok, &ds, nullptr))) return rc;
            after[0] = avg[0]; after[1] = avg[1]; after[2] = avg[2];
            mcx_avg_walk(after, ok, ds, nc, est.data());
            if (tail && iter == 0 && memcmp(est.data(), c->tail.h + 18 + 3 * nc, (size_t)nc * 4) == 0) {
                // the device walked the same trajectory: its list of the pairs whose estimate moved is the list
                n_redo = c->tail.h[0];
                const auto tq = std::chrono::steady_clock::now();
                if (n_redo && (rc = replay_listed(c, n_redo, stats))) return rc;
                if (n_redo && c->kn.timing) fprintf(stderr, "[mcx] avgDist: %u pairs of the batch re-run (the device's list), %.2f ms\n", n_redo, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq).count());
                if (n_redo == 0) break;
                continue;
            }
            if (tail && iter == 0 && c->kn.timing) fprintf(stderr, "[mcx] the device's walk of the batch's chunks differs from the host's: the host's is taken\n");
            const auto tq = std::chrono::steady_clock::now();
            if ((rc = mcx_batch_replay(c, est.data(), &n_redo, stats))) return rc;
            if (c->kn.timing && (n_redo || iter)) fprintf(stderr, "[mcx] avgDist: pass %d, %u pairs re-run, %.2f ms\n", iter, n_redo, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq).count());
            if (n_redo == 0) break;
            if (iter == 63) return fail(MCX_ERR_CAPACITY, "avgDist replay did not converge");
        }
    }
    const auto t_loop = std::chrono::steady_clock::now();
    if ((rc = mcx_batch_end(c, stats))) return rc;
    if (c->kn.timing) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto t_out = std::chrono::steady_clock::now();
        if (ms(t_in, t_out) > 30) fprintf(stderr, "[mcx] a slow batch (%u reads): set-up %.2f ms, first pass %.2f ms, avgDist %.2f ms, closing %.2f ms\n", n_reads, c->run.ms_setup, ms(t_in, c->run.t_begun) - c->run.ms_setup, ms(c->run.t_begun, t_loop), ms(t_loop, t_out));
    }
    avg[0] = after[0]; avg[1] = after[1]; avg[2] = after[2];
    avg[3] += n_reads;
    return 0;
}

// host buffers -> the context's staging arrays in HBM (what mcx_map_batch and the file front end hand the step API)
int mcx_stage_in(mcx_ctx *c, const uint8_t *bases, const uint32_t *off, uint32_t n_reads, const uint8_t **d_bases, const uint32_t **d_off,
                 mcx_aln **d_aln, uint32_t **d_cigar)
{
    if (n_reads > c->max_reads) return fail(MCX_ERR_ARG, "batch larger than max_batch_reads");
    HIP_TRY(hipSetDevice(c->idx->device));
    int rc;
    if (!c->d_bases) {
        if ((rc = dmalloc(&c->d_bases, c->max_bases + 64))) return rc;
        if ((rc = dmalloc(&c->d_off, c->max_reads + 1))) return rc;
        if ((rc = dmalloc(&c->d_recs, c->max_reads))) return rc;
        if ((rc = dmalloc(&c->d_cig, MCX_CIGAR_POOL_WORDS(c->max_reads)))) return rc;
    }
    if (off[n_reads] > c->max_bases) return fail(MCX_ERR_ARG, "batch holds more bases than max_batch_reads * max_read_len");
    HIP_TRY(hipMemcpyAsync(c->d_bases, bases, off[n_reads], hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_off, off, (size_t)(n_reads + 1) * 4, hipMemcpyHostToDevice, c->stream));
    *d_bases = c->d_bases; *d_off = c->d_off; *d_aln = (mcx_aln *)c->d_recs; *d_cigar = c->d_cig;
    return 0;
}

int mcx_stage_out(mcx_ctx *c, uint32_t n_reads, mcx_aln *aln, uint32_t *cigar)
{
    HIP_TRY(hipSetDevice(c->idx->device));
    HIP_TRY(hipMemcpyAsync(aln, c->d_recs, (size_t)n_reads * sizeof(AlnRec), hipMemcpyDeviceToHost, c->stream));
    if (c->run.cig_words) HIP_TRY(hipMemcpyAsync(cigar, c->d_cig, (size_t)c->run.cig_words * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int mcx_map_batch(mcx_ctx *c, const uint8_t *bases, const uint32_t *off, uint32_t n_reads, int paired,
                             int64_t avg[4], mcx_aln *aln, uint32_t *cigar, mcx_stats *stats)
{
    if (!c || !bases || !off || !aln || !cigar) return fail(MCX_ERR_ARG, "mcx_map_batch: null argument");
    if (n_reads == 0) return 0;
    const uint8_t *d_bases; const uint32_t *d_off; mcx_aln *d_aln; uint32_t *d_cig;
    int rc = mcx_stage_in(c, bases, off, n_reads, &d_bases, &d_off, &d_aln, &d_cig);
    if (rc) return rc;
    if ((rc = mcx_map_batch_dev(c, d_bases, d_off, n_reads, paired, avg, d_aln, d_cig, stats))) return rc;
    return mcx_stage_out(c, n_reads, aln, cigar);
}

// ---------------------------------------------------------------------------------------------
// batches from host memory with the copies overlapped with the kernels
// ---------------------------------------------------------------------------------------------
// Bulk copies across the device boundary: the DMA engines by default (52 GB/s each way on the test box, and they leave the
// CUs to the kernels).  MCX_STREAM_KERNEL_COPY=1 moves them with a kernel instead (page-locked host memory is mapped into
// the device's address space) — measured slower next to the mapping kernels (82 ms instead of 70 ms per 8 M-read batch), kept for
// boxes whose DMA queues are the bottleneck.
__global__ void __launch_bounds__(256) k_copy16(const U4 *__restrict__ src, U4 *__restrict__ dst, uint64_t n16)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static int bulk_copy(mcx_ctx *c, void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s)
{
    if (bytes == 0) return 0;
    // (MCX_STREAM_KERNEL_COPY: this library's own copy kernel over the mapped host memory instead of the runtime's copies — round 5 tried it for the way out
    //  alone, on grids of 16 / 48 / 128 workgroups, next to the mapping kernels: 23.7 / 26.1 / 27.2 ms per step against the runtime's 20.6)
    static const char *mode = getenv("MCX_STREAM_KERNEL_COPY"); // ("in" / "out": one direction alone)
    static const int blocks = getenv("MCX_COPY_BLOCKS") ? atoi(getenv("MCX_COPY_BLOCKS")) : 512;
    const bool use_dma = mode == nullptr || (!strcmp(mode, "in") && kind != hipMemcpyHostToDevice) || (!strcmp(mode, "out") && kind != hipMemcpyDeviceToHost);
    const size_t n16 = bytes / 16;
    bool mapped = false; // is the host side page-locked memory the device can address?
    {
        hipPointerAttribute_t a;
        const void *host = kind == hipMemcpyHostToDevice ? src : dst;
        if (hipPointerGetAttributes(&a, host) == hipSuccess) mapped = a.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }
    if (use_dma || !mapped || n16 == 0 || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) { HIP_TRY(hipMemcpyAsync(dst, src, bytes, kind, s)); return 0; }
    k_copy16<<<blocks, 256, 0, s>>>((const U4 *)src, (U4 *)dst, (uint64_t)n16);
    HIP_TRY(hipGetLastError());
    if (bytes & 15) HIP_TRY(hipMemcpyAsync((uint8_t *)dst + n16 * 16, (const uint8_t *)src + n16 * 16, bytes & 15, kind, s));
    (void)c;
    return 0;
}

static mcx_ctx::Slot *oldest_slot(mcx_ctx *c, int state)
{
    mcx_ctx::Slot *best = nullptr;
    for (auto &sl : c->slot) if (sl.state == state && (!best || sl.seq < best->seq)) best = &sl;
    return best;
}

// a free slot, its HBM allocated on first use
static int stream_slot(mcx_ctx *c, mcx_ctx::Slot **out)
{
    mcx_ctx::Slot *sl = oldest_slot(c, 0);
    if (!sl) return fail(MCX_ERR_ARG, "mcx_stream_submit: three batches are in flight (collect one first)");
    int rc;
    if (!c->h2d_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->h2d_stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
    }
    if (!sl->d_bases) {
        if ((rc = dmalloc(&sl->d_bases, c->max_bases + 16 * c->max_reads + 64))) return rc; // (+16 per read: packed rows end on a word)
        if ((rc = dmalloc(&sl->d_off, c->max_reads + 1))) return rc;
        if ((rc = dmalloc(&sl->d_recs, c->max_reads))) return rc;
        if ((rc = dmalloc(&sl->d_cig, MCX_CIGAR_POOL_WORDS(c->max_reads)))) return rc;
        HIP_TRY(hipEventCreateWithFlags(&sl->in_ready, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl->mapped, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl->out_done, hipEventDisableTiming));
    }
    *out = sl;
    return 0;
}

// 2-bit rows -> the ASCII bytes of the batch: one thread per sixteen bases (the letters ACGT; k_apply_odd puts back every other byte)
// (the lengths are the caller's: one beyond its row or the context's longest read — lim —, or a sum beyond the slot — max_bases —, is flagged in
//  *err and nothing is written for it; mcx_stream_next refuses the batch.  Round 4 walked the lengths on the host before the copy: a
//  millisecond per 8 M reads with the GPU idle.)
__global__ void __launch_bounds__(256) k_unpack_reads(const uint32_t *codes, uint32_t row_words, const uint32_t *off, uint32_t n_reads, uint8_t *bases,
                                                      uint64_t max_bases, uint32_t lim, uint32_t *err)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = (uint32_t)(t / row_words), k = (uint32_t)(t % row_words);
    if (r >= n_reads) return;
    const uint32_t o = off[r], rlen = off[r + 1] - o;
    if (rlen > lim || off[r + 1] < o) { if (k == 0) atomicOr(err, 1u); return; }
    if ((uint64_t)o + rlen > max_bases) { if (k == 0) atomicOr(err, 2u); return; }
    if (16 * k >= rlen) return;
    const uint32_t w = codes[(uint64_t)r * row_words + k];
    uint32_t q[4]; // sixteen letters, four to a word, the first in the low byte
#pragma unroll
    for (int g = 0; g < 4; g++) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t c = (w >> (30 - 2 * (4 * g + i))) & 3u;
            v |= (uint32_t)((0x54474341u >> (8 * c)) & 0xFFu) << (8 * i); // "ACGT"
        }
        q[g] = v;
    }
    uint8_t *dst = bases + o + 16 * k;
    const uint32_t nb = rlen - 16 * k < 16 ? rlen - 16 * k : 16;
    const uintptr_t a = (uintptr_t)dst;
    if (nb == 16 && (a & 3) == 0) { uint32_t *d4 = (uint32_t *)dst; d4[0] = q[0]; d4[1] = q[1]; d4[2] = q[2]; d4[3] = q[3]; }
    else if (nb == 16 && (a & 1) == 0) { uint16_t *d2 = (uint16_t *)dst; for (int i = 0; i < 8; i++) d2[i] = (uint16_t)(q[i >> 1] >> (16 * (i & 1))); }
    else for (uint32_t i = 0; i < nb; i++) dst[i] = (uint8_t)(q[i >> 2] >> (8 * (i & 3)));
}

__global__ void k_apply_odd(const uint64_t *odd, uint32_t n_odd, const uint32_t *off, uint32_t n_reads, uint8_t *bases, uint64_t max_bases)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_odd) return;
    const uint64_t e = odd[i];
    const uint32_t r = (uint32_t)(e >> 32), pos = (uint32_t)(e >> 8) & 0xFFFFFFu;
    if (r < n_reads && off[r + 1] >= off[r] && pos < off[r + 1] - off[r] && (uint64_t)off[r] + pos < max_bases) bases[off[r] + pos] = (uint8_t)e;
}

// a batch whose lengths k_unpack_reads refused maps nothing: every read becomes empty, so that no kernel behind this one meets a length it was not sized
// for — the host hears of it when it next looks (mcx_stream_map), not before the batch's first kernel: no wait at the start of a step
__global__ void __launch_bounds__(256) k_neutralize(uint32_t *off, uint32_t n_reads, const uint32_t *err)
{
    if (*err == 0) return;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r <= n_reads; r += gridDim.x * blockDim.x) off[r] = 0;
}

extern "C" int mcx_stream_submit_packed(mcx_ctx *c, const uint32_t *codes, uint32_t row_words, const uint32_t *len, uint32_t n_reads, const uint64_t *odd,
                                        uint32_t n_odd)
{
    if (!c || !codes || !len || n_reads == 0 || row_words == 0 || (n_odd && !odd)) return fail(MCX_ERR_ARG, "mcx_stream_submit_packed: bad argument");
    if (n_reads > c->max_reads) return fail(MCX_ERR_ARG, "batch larger than max_batch_reads");
    const uint32_t row_max = (uint32_t)(c->rlen_max + 15) / 16;
    if (row_words > row_max) return fail(MCX_ERR_UNSUPPORTED, "mcx_stream_submit_packed: rows are longer than max_read_len");
    HIP_TRY(hipSetDevice(c->idx->device));
    mcx_ctx::Slot *sl = nullptr;
    int rc = stream_slot(c, &sl);
    if (rc) return rc;
    hipStream_t s = c->h2d_stream;
    if (!sl->d_codes) {
        if ((rc = dmalloc(&sl->d_codes, c->max_reads * (uint64_t)row_max))) return rc;
        if ((rc = dmalloc(&sl->d_len, c->max_reads + 1))) return rc;
        if ((rc = dmalloc(&sl->d_err, 1))) return rc;
        HIP_TRY(hipHostMalloc((void **)&sl->h_err, sizeof(uint32_t)));
        *sl->h_err = 0;
    }
    if (n_odd > sl->odd_cap) {
        if (sl->d_odd) { HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(sl->d_odd); sl->d_odd = nullptr; }
        sl->odd_cap = std::max<uint32_t>(n_odd + n_odd / 2, 1u << 16);
        if ((rc = dmalloc(&sl->d_odd, sl->odd_cap))) return rc;
    }
    if (!c->d_scan_tmp) {
        size_t need = 0;
        HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, need, sl->d_len, sl->d_off + 1, (int)c->max_reads, s));
        c->scan_tmp_bytes = need + 256;
        HIP_TRY(hipMalloc(&c->d_scan_tmp, c->scan_tmp_bytes));
    }
    if ((rc = bulk_copy(c, sl->d_codes, codes, (size_t)n_reads * row_words * 4, hipMemcpyHostToDevice, s))) return rc;
    if ((rc = bulk_copy(c, sl->d_len, len, (size_t)n_reads * 4, hipMemcpyHostToDevice, s))) return rc;
    if (n_odd && (rc = bulk_copy(c, sl->d_odd, odd, (size_t)n_odd * 8, hipMemcpyHostToDevice, s))) return rc;
    HIP_TRY(hipMemsetAsync(sl->d_off, 0, 4, s));
    size_t tmp = c->scan_tmp_bytes;
    HIP_TRY(hipcub::DeviceScan::InclusiveSum(c->d_scan_tmp, tmp, sl->d_len, sl->d_off + 1, (int)n_reads, s));
    const uint64_t threads = (uint64_t)n_reads * row_words;
    HIP_TRY(hipMemsetAsync(sl->d_err, 0, 4, s));
    k_unpack_reads<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(sl->d_codes, row_words, sl->d_off, n_reads, sl->d_bases, c->max_bases,
                                                                   std::min<uint32_t>(row_words * 16u, (uint32_t)c->rlen_max), sl->d_err);
    if (n_odd) k_apply_odd<<<(n_odd + 255) / 256, 256, 0, s>>>(sl->d_odd, n_odd, sl->d_off, n_reads, sl->d_bases, c->max_bases);
    k_neutralize<<<256, 256, 0, s>>>(sl->d_off, n_reads, sl->d_err);
    HIP_TRY(hipGetLastError());
    sl->lens_checked = true;
    // the batch's 2-bit form for the kernels, made here — behind its copy in, under the batch before it — instead of at the start of its own step (0.76 ms of
    // the step per 8 M reads).  Mated or not is a guess (what the last batch was); a wrong one, or a profile attached meanwhile, and the step packs as before.
    sl->prepacked = false;
    if (!c->prof_planes && !c->kn.no_prepack) {
        if (!sl->d_prepack) {
            if ((rc = dmalloc(&sl->d_prepack, c->max_reads * (uint64_t)c->wpad))) return rc;
            if ((rc = dmalloc(&sl->d_any_n, 1))) return rc;
        }
        ReadBatch rb; rb.bases = sl->d_bases; rb.off = sl->d_off; rb.n_reads = n_reads;
        const int tpr = (c->rlen_max + 31) / 32 + 1;
        HIP_TRY(hipMemsetAsync(sl->d_any_n, 0, 4, s));
        k_pack_reads<<<(unsigned)(((uint64_t)n_reads * (uint64_t)tpr + 255) / 256), 256, 0, s>>>(rb, c->last_paired, c->wpad, tpr, sl->d_prepack, sl->d_any_n, nullptr);
        HIP_TRY(hipGetLastError());
        sl->prepacked = true; sl->pre_paired = c->last_paired;
    }
    HIP_TRY(hipEventRecord(sl->in_ready, s));
    sl->n_reads = n_reads; sl->state = 1; sl->seq = ++c->stream_seq;
    c->stream_bytes_in += (uint64_t)n_reads * row_words * 4 + (uint64_t)n_reads * 4 + (uint64_t)n_odd * 8;
    return 0;
}

extern "C" int mcx_stream_submit(mcx_ctx *c, const uint8_t *bases, const uint32_t *off, uint32_t n_reads)
{
    if (!c || !bases || !off || n_reads == 0) return fail(MCX_ERR_ARG, "mcx_stream_submit: bad argument");
    if (n_reads > c->max_reads) return fail(MCX_ERR_ARG, "batch larger than max_batch_reads");
    if (off[n_reads] > c->max_bases) return fail(MCX_ERR_ARG, "batch holds more bases than max_batch_reads * max_read_len");
    HIP_TRY(hipSetDevice(c->idx->device));
    mcx_ctx::Slot *sl = nullptr;
    int rc = stream_slot(c, &sl);
    if (rc) return rc;
    if (sl->d_err) HIP_TRY(hipMemsetAsync(sl->d_err, 0, 4, c->h2d_stream)); // (the slot once took 2-bit rows: nothing of that batch's verdict is this one's)
    sl->lens_checked = false; sl->prepacked = false;
    if ((rc = bulk_copy(c, sl->d_bases, bases, off[n_reads], hipMemcpyHostToDevice, c->h2d_stream))) return rc;
    if ((rc = bulk_copy(c, sl->d_off, off, (size_t)(n_reads + 1) * 4, hipMemcpyHostToDevice, c->h2d_stream))) return rc;
    HIP_TRY(hipEventRecord(sl->in_ready, c->h2d_stream));
    sl->n_reads = n_reads; sl->state = 1; sl->seq = ++c->stream_seq;
    c->stream_bytes_in += (uint64_t)off[n_reads] + (uint64_t)(n_reads + 1) * 4;
    return 0;
}

// the oldest submitted batch, in HBM once the context's stream gets there
extern "C" int mcx_stream_next(mcx_ctx *c, const uint8_t **d_bases, const uint32_t **d_off, uint32_t *n_reads, mcx_aln **d_aln, uint32_t **d_cigar)
{
    if (!c || !d_bases || !d_off || !d_aln || !d_cigar) return fail(MCX_ERR_ARG, "mcx_stream_next: null argument");
    HIP_TRY(hipSetDevice(c->idx->device));
    if (oldest_slot(c, 2)) return fail(MCX_ERR_ARG, "mcx_stream_next: the previous batch was not handed back (mcx_stream_mapped)");
    mcx_ctx::Slot *sl = oldest_slot(c, 1);
    if (!sl) return fail(MCX_ERR_ARG, "mcx_stream_next: nothing submitted");
    HIP_TRY(hipStreamWaitEvent(c->stream, sl->in_ready, 0));
    c->lens_checked = sl->lens_checked; // (for the mcx_batch_begin that follows: no need to look for an over-long read, nor to wait for the answer)
    c->lens_checked_off = sl->d_off; c->lens_checked_bases = sl->d_bases;
    c->pre = mcx_ctx::PrePacked();
    if (sl->prepacked) { c->pre.packed = sl->d_prepack; c->pre.bases = sl->d_bases; c->pre.paired = sl->pre_paired; c->pre.any_n = sl->d_any_n; }
    sl->state = 2;
    *d_bases = sl->d_bases; *d_off = sl->d_off; *d_aln = (mcx_aln *)sl->d_recs; *d_cigar = sl->d_cig;
    if (n_reads) *n_reads = sl->n_reads;
    return 0;
}

// the batch mcx_stream_next gave out is mapped: its results start their way to host memory
extern "C" int mcx_stream_mapped(mcx_ctx *c, mcx_aln *aln, uint32_t *cigar)
{
    if (!c || !aln || !cigar) return fail(MCX_ERR_ARG, "mcx_stream_mapped: null argument");
    HIP_TRY(hipSetDevice(c->idx->device));
    mcx_ctx::Slot *sl = oldest_slot(c, 2);
    if (!sl) return fail(MCX_ERR_ARG, "mcx_stream_mapped: no batch is being mapped");
    HIP_TRY(hipEventRecord(sl->mapped, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->d2h_stream, sl->mapped, 0));
    const size_t rec_bytes = (size_t)sl->n_reads * sizeof(AlnRec), cig_bytes = (size_t)c->run.cig_words * 4; // (the pool's used words only)
    int rc;
    if ((rc = bulk_copy(c, aln, sl->d_recs, rec_bytes, hipMemcpyDeviceToHost, c->d2h_stream))) return rc;
    if ((rc = bulk_copy(c, cigar, sl->d_cig, cig_bytes, hipMemcpyDeviceToHost, c->d2h_stream))) return rc;
    if (sl->lens_checked) HIP_TRY(hipMemcpyAsync(sl->h_err, sl->d_err, 4, hipMemcpyDeviceToHost, c->d2h_stream)); // what k_unpack_reads thought of the caller's lengths: mcx_stream_collect reads it
    HIP_TRY(hipEventRecord(sl->out_done, c->d2h_stream));
    sl->state = 3;
    c->stream_bytes_out += rec_bytes + cig_bytes;
    return 0;
}

// the records of a mapped batch in 32 bytes each (mcx_aln32, include/mcx.h) for their way to the host
__global__ void __launch_bounds__(256) k_pack_recs(const AlnRec *recs, uint32_t n, mcx_aln32 *out)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const AlnRec a = recs[r];
    mcx_aln32 o;
    o.pos_lo = (uint32_t)a.pos; o.pos_hi = (uint8_t)((uint64_t)a.pos >> 32); o.mate_lo = (uint32_t)a.mate_pos; o.mate_hi = (uint8_t)((uint64_t)a.mate_pos >> 32);
    o.mapq = (uint8_t)a.mapq; o.bits = (uint8_t)((a.fwd ? 1 : 0) | (a.has_mate ? 2 : 0)); o.tlen = a.tlen; o.flag = (uint16_t)a.flag;
    o.chr = a.chr < 0 ? (uint16_t)0xFFFFu : (uint16_t)a.chr; o.nm = (int16_t)a.nm; o.as = (int16_t)a.as; o.xs = (int16_t)a.xs;
    o.n_cigar = (uint16_t)a.n_cigar; o.cigar_off = (uint32_t)a.pad[0];
    ((U4 *)out)[2 * (uint64_t)r] = ((const U4 *)&o)[0]; ((U4 *)out)[2 * (uint64_t)r + 1] = ((const U4 *)&o)[1];
}

extern "C" int mcx_stream_mapped32(mcx_ctx *c, mcx_aln32 *aln, uint32_t *cigar)
{
    static_assert(sizeof(mcx_aln32) == 32, "mcx_aln32 is two 16-byte words");
    if (!c || !aln || !cigar) return fail(MCX_ERR_ARG, "mcx_stream_mapped32: null argument");
    HIP_TRY(hipSetDevice(c->idx->device));
    mcx_ctx::Slot *sl = oldest_slot(c, 2);
    if (!sl) return fail(MCX_ERR_ARG, "mcx_stream_mapped32: no batch is being mapped");
    if (c->idx->view.n_chr >= 0xFFFF || (c->idx->view.G2 >> 40)) return fail(MCX_ERR_UNSUPPORTED, "mcx_stream_mapped32: more than 65534 contigs or positions beyond 2^40 (use mcx_stream_mapped)");
    int rc;
    if (!sl->d_recs32 && (rc = dmalloc(&sl->d_recs32, c->max_reads))) return rc;
    k_pack_recs<<<(sl->n_reads + 255) / 256, 256, 0, c->stream>>>(sl->d_recs, sl->n_reads, sl->d_recs32); // (a read's operations are at most MCX_CIGAR_STRIDE x the pool's slack: far below 2^16)
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sl->mapped, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->d2h_stream, sl->mapped, 0));
    const size_t rec_bytes = (size_t)sl->n_reads * sizeof(mcx_aln32), cig_bytes = (size_t)c->run.cig_words * 4; // (the pool's used words only)
    if ((rc = bulk_copy(c, aln, sl->d_recs32, rec_bytes, hipMemcpyDeviceToHost, c->d2h_stream))) return rc;
    if ((rc = bulk_copy(c, cigar, sl->d_cig, cig_bytes, hipMemcpyDeviceToHost, c->d2h_stream))) return rc;
    if (sl->lens_checked) HIP_TRY(hipMemcpyAsync(sl->h_err, sl->d_err, 4, hipMemcpyDeviceToHost, c->d2h_stream));
    HIP_TRY(hipEventRecord(sl->out_done, c->d2h_stream));
    sl->state = 3;
    c->stream_bytes_out += rec_bytes + cig_bytes;
    return 0;
}

static int stream_map(mcx_ctx *c, int paired, int64_t avg[4], mcx_aln *aln, mcx_aln32 *aln32, uint32_t *cigar, mcx_stats *stats);
extern "C" int mcx_stream_map32(mcx_ctx *c, int paired, int64_t avg[4], mcx_aln32 *aln, uint32_t *cigar, mcx_stats *stats) { return stream_map(c, paired, avg, nullptr, aln, cigar, stats); }
extern "C" int mcx_stream_map(mcx_ctx *c, int paired, int64_t avg[4], mcx_aln *aln, uint32_t *cigar, mcx_stats *stats) { return stream_map(c, paired, avg, aln, nullptr, cigar, stats); }
static int stream_map(mcx_ctx *c, int paired, int64_t avg[4], mcx_aln *aln, mcx_aln32 *aln32, uint32_t *cigar, mcx_stats *stats)
{
    const uint8_t *d_bases; const uint32_t *d_off; mcx_aln *d_aln; uint32_t *d_cig; uint32_t n = 0;
    int rc = mcx_stream_next(c, &d_bases, &d_off, &n, &d_aln, &d_cig);
    if (rc) return rc;
    rc = mcx_map_batch_dev(c, d_bases, d_off, n, paired, avg, d_aln, d_cig, stats);
    mcx_ctx::Slot *sl = oldest_slot(c, 2);
    if (rc == 0 && sl->lens_checked) { // what k_unpack_reads thought of the caller's lengths (a refused batch was mapped as empty reads)
        uint32_t err = 0;
        HIP_TRY(hipMemcpy(&err, sl->d_err, 4, hipMemcpyDeviceToHost));
        if (err) rc = fail(MCX_ERR_ARG, (err & 1u) ? "mcx_stream_submit_packed: a read is longer than its row / max_read_len" : "batch holds more bases than max_batch_reads * max_read_len");
    }
    if (rc) { sl->state = 0; return rc; }
    return aln32 ? mcx_stream_mapped32(c, aln32, cigar) : mcx_stream_mapped(c, aln, cigar);
}

extern "C" int mcx_stream_collect(mcx_ctx *c, uint64_t *bytes_in, uint64_t *bytes_out)
{
    if (!c) return fail(MCX_ERR_ARG, "mcx_stream_collect: null argument");
    HIP_TRY(hipSetDevice(c->idx->device));
    mcx_ctx::Slot *sl = oldest_slot(c, 3);
    if (!sl) return fail(MCX_ERR_ARG, "mcx_stream_collect: no mapped batch is on its way out");
    HIP_TRY(hipEventSynchronize(sl->out_done));
    sl->state = 0;
    if (bytes_in) *bytes_in = c->stream_bytes_in;
    if (bytes_out) *bytes_out = c->stream_bytes_out;
    if (sl->lens_checked && sl->h_err && *sl->h_err) { // the two-half form (mcx_stream_next + mcx_map_batch_dev / mcx_batch_* + mcx_stream_mapped): the refusal arrives with the records
        const uint32_t err = *sl->h_err;
        *sl->h_err = 0;
        return fail(MCX_ERR_ARG, (err & 1u) ? "mcx_stream_submit_packed: a read is longer than its row / max_read_len (the batch was mapped as empty reads)"
                                            : "mcx_stream_submit_packed: the batch holds more bases than max_batch_reads * max_read_len (it was mapped as empty reads)");
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// -vcf bookkeeping: UpdateProfile / UpdateMultiHitCount for a finished batch (mcx_profile.h)
// ---------------------------------------------------------------------------------------------
// discordant-pair events, ReadMapping.cpp:486-521: kept as seen ('E' records: pos = the pair's number in the input stream,
// len = kind, seq = g1, g2, dist) — the reference's second branch pushes its DiscordPair variable whatever the previous
// discordant pair left in it, so they are resolved in input order once the run is over
// (mcx_disc_resolve: mcx_profile_sparse for one shard, mcx_call_variants for several)
__global__ void k_prof_disc(const uint8_t *detail, DetailLayout dl, uint32_t n_pairs, int64_t first_pair, SparseRec *out, uint32_t *n, uint32_t cap)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const DetailHdr &d = *(const DetailHdr *)(detail + (uint64_t)(2 * p) * dl.stride);
    if (d.disc_kind == 0) return;
    const uint32_t at = atomicAdd(n, 1u);
    if (at >= cap) return;
    SparseRec e;
    uint64_t *w = (uint64_t *)&e;
    for (int k = 0; k < (int)(sizeof(SparseRec) / 8); k++) w[k] = 0;
    e.pos = first_pair + p; e.type = 'E'; e.len = (uint8_t)d.disc_kind;
    const int64_t v[3] = {d.disc_g1, d.disc_g2, d.disc_dist};
    memcpy(e.seq, v, sizeof v);
    out[at] = e;
}

extern "C" int mcx_profile_attach(mcx_ctx *c, uint32_t *d_planes, int max_dup, int max_clip)
{
    static_assert(sizeof(mcx_sparse_rec) == sizeof(SparseRec), "mcx_sparse_rec and SparseRec must have one layout");
    if (!c || !d_planes) return fail(MCX_ERR_ARG, "mcx_profile_attach: null argument");
    if (c->idx->view.G >= (int64_t)1 << 32) return fail(MCX_ERR_UNSUPPORTED, "the alignment profile takes genomes below 2^32 bases");
    HIP_TRY(hipSetDevice(c->idx->device));
    if (c->later.pending) { (void)hipStreamSynchronize(c->later.stream); c->later.pending = false; } // (a profile that is given up with a batch's bookkeeping on its way)
    c->prof_planes = d_planes;
    c->prof_max_dup = (max_dup <= 0 || max_dup > 15) ? 15 : max_dup; // main.cpp:240-244, :323
    c->prof_max_clip = max_clip;
    if (!c->d_detail) {
        c->dlay = make_detail_layout(c->rlen_max);
        int rc;
        if ((rc = dmalloc(&c->d_detail, (size_t)c->dlay.stride * c->max_reads))) return rc;
        if ((rc = dmalloc(&c->d_admit, c->max_reads + 4))) return rc;
        c->prof_items_cap = (uint32_t)std::min<uint64_t>((uint64_t)c->max_reads * 4 + 1024, 0x7fffffffu);
        if ((rc = dmalloc(&c->d_prof_items, c->prof_items_cap))) return rc;
        if ((rc = sort_reserve(c, c->max_reads))) return rc;
        c->sparse_cap = (uint32_t)std::min<uint64_t>(c->max_reads * 2 + 4096, 0x7fffffffu);
        if ((rc = dmalloc(&c->d_sparse, c->sparse_cap))) return rc;
        HIP_TRY(hipHostMalloc((void **)&c->h_sparse_pin, (size_t)c->sparse_pin_recs * sizeof(SparseRec)));
    }
    c->h_sparse.clear(); c->h_events.clear(); c->n_tally = 0; c->arch.n = c->arch_ev.n = 0;
    c->arch.host = &c->h_sparse; c->arch_ev.host = &c->h_events;
    if (!c->arch.d) { // room for the first batches' records now, not in the middle of the first batch
        const uint64_t first = std::min<uint64_t>(std::max<uint64_t>(c->sparse_cap, (uint64_t)1 << 20), (uint64_t)1 << 22);
        if (hipMalloc((void **)&c->arch.d, first * sizeof(SparseRec)) == hipSuccess) c->arch.cap = first; else { (void)hipGetLastError(); c->arch.d = nullptr; }
    }
    const size_t match_n = (size_t)planes_stride(c->idx->view.G);
    if (!c->d_prof_match) {
        int rc = dmalloc(&c->d_prof_match, match_n);
        if (rc && c->later.have) { // the device has filled up since the second set was taken: it goes back, the bookkeeping runs inside the batches' calls from here on
            auto &L = c->later;
            (void)hipGetLastError();
            (void)hipStreamSynchronize(L.stream); (void)hipStreamSynchronize(L.keep_stream);
            for (void **q : {(void **)&L.d_detail_alt, (void **)&L.d_admit_alt, (void **)&L.d_keep_bases, (void **)&L.d_keep_off, (void **)&L.d_keep_bases_alt, (void **)&L.d_keep_off_alt, (void **)&L.d_ev}) if (*q) { (void)hipFree(*q); *q = nullptr; }
            L.have = false;
            if (c->kn.timing) fprintf(stderr, "[mcx profile] the second set of detail records goes back: the device has no room for the coverage plane beside it\n");
            rc = dmalloc(&c->d_prof_match, match_n);
        }
        if (rc) return rc;
    }
    HIP_TRY(hipMemsetAsync(c->d_prof_match, 0, match_n * sizeof(uint16_t), c->stream));
    if (!c->kn.no_prof_overlap && !c->later.have && !c->later.tried) { // a second set of what a batch's mapping writes for the bookkeeping, if HBM has the room (without it: the bookkeeping inside the call, as before)
        auto &L = c->later;
        L.tried = true;
        L.ev_cap = (uint32_t)std::min<uint64_t>(c->max_reads / 2 + 4096, 0x7fffffffu);
        // (a stream of the lowest priority: queues of a priority of their own in the runtime — a stream of the default priority may share a hardware queue with the
        //  context's main stream, and then the bookkeeping runs before the next batch's kernels, not under them — and the mapping goes first where both want the chip)
        int prio_low = 0, prio_high = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
        size_t hbm_free = 0, hbm_all = 0;
        (void)hipMemGetInfo(&hbm_free, &hbm_all);
        const size_t second_set = (size_t)c->dlay.stride * c->max_reads + 2 * (c->max_bases + 4 * c->max_reads) + (size_t)L.ev_cap * sizeof(SparseRec);
        // (what is left afterwards has to hold the record archive's growth and the caller's own buffers: four gigabytes at least)
        bool ok = hbm_free > second_set + ((size_t)4 << 30) && hipMalloc((void **)&L.d_detail_alt, (size_t)c->dlay.stride * c->max_reads) == hipSuccess && hipMalloc((void **)&L.d_admit_alt, c->max_reads + 4) == hipSuccess &&
                  hipMalloc((void **)&L.d_keep_bases, c->max_bases + 64) == hipSuccess && hipMalloc((void **)&L.d_keep_off, (c->max_reads + 1) * sizeof(uint32_t)) == hipSuccess &&
                  hipMalloc((void **)&L.d_keep_bases_alt, c->max_bases + 64) == hipSuccess && hipMalloc((void **)&L.d_keep_off_alt, (c->max_reads + 1) * sizeof(uint32_t)) == hipSuccess &&
                  hipMalloc((void **)&L.d_cnt, CNT_N * sizeof(uint32_t)) == hipSuccess && hipMalloc((void **)&L.d_ev, (size_t)L.ev_cap * sizeof(SparseRec)) == hipSuccess &&
                  hipHostMalloc((void **)&L.h_cnt, CNT_N * sizeof(uint32_t)) == hipSuccess &&
                  hipStreamCreateWithPriority(&L.stream, hipStreamNonBlocking, prio_low) == hipSuccess && hipStreamCreateWithPriority(&L.keep_stream, hipStreamNonBlocking, prio_low) == hipSuccess;
        for (hipEvent_t *e : {&L.go, &L.done, &L.kept, &L.begun}) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            for (void **q : {(void **)&L.d_detail_alt, (void **)&L.d_admit_alt, (void **)&L.d_keep_bases, (void **)&L.d_keep_off, (void **)&L.d_keep_bases_alt, (void **)&L.d_keep_off_alt, (void **)&L.d_cnt, (void **)&L.d_ev}) if (*q) { (void)hipFree(*q); *q = nullptr; }
            if (c->kn.timing) fprintf(stderr, "[mcx profile] no room in HBM for a second set of detail records: a batch's bookkeeping runs inside its call\n");
        }
        L.have = ok;
    }
    c->prof_settled = false; c->prof_broken = false;
    return 0;
}

// A context — and, for a -vcf run, the planes and the bookkeeping's buffers — sized to what the device has left (the -vcf leg at 3.1 Gbp left
// 5.4 GB of 309, and a larger genome met a bare hipMalloc failure in the middle of its first batch).  Everything the run will take is taken HERE: the context,
// tier 0's pair records (otherwise the first batch's), the planes, the detail records.  When that does not fit with kFitMargin to spare — the record
// archive's growth, the caller's own batches, the stream slots — the run is degraded in a fixed order, each step said on stderr in one line:
//   1. the index gives its pair records back (mcx_index_trim: the seeding walk takes one base per step, 24.8 GB at 3.1 Gbp);
//   2. max_batch_reads is halved, again and again (down to 128 K reads);
// (a second set of detail records, which lets a batch's bookkeeping run under the next batch, is only ever taken when there is room: mcx_profile_attach).
// MCX_HBM_CAP_GB=n (tests): the run may take n GB, whatever the device has free.
constexpr size_t kFitMargin = (size_t)4 << 30;
extern "C" int mcx_ctx_create_fit(mcx_index *ix, const mcx_opts *opts, int with_profile, int paired, int max_dup, int max_clip, mcx_ctx **out, uint32_t **planes, mcx_fit *fit)
{
    if (!ix || !out || (with_profile && !planes)) return fail(MCX_ERR_ARG, "mcx_ctx_create_fit: null argument");
    mcx_opts o;
    if (opts) o = *opts; else mcx_opts_default(&o);
    mcx_fit f; memset(&f, 0, sizeof f);
    size_t cap = 0; // MCX_HBM_CAP_GB=n (tests): the run may take n GB of the device, whatever is free
    if (const char *e = getenv("MCX_HBM_CAP_GB")) cap = (size_t)(atof(e) * (double)((size_t)1 << 30));
    HIP_TRY(hipSetDevice(ix->device));
    for (;;) {
        mcx_ctx *c = nullptr;
        uint32_t *pl = nullptr;
        size_t free0 = 0, hbm_free = 0, hbm_all = 0;
        (void)hipMemGetInfo(&free0, &hbm_all);
        int rc = mcx_ctx_create(ix, &o, &c);
        if (rc == 0) rc = reserve_tier0(c, paired, (uint64_t)o.max_batch_reads);
        if (rc == 0 && with_profile) {
            rc = mcx_planes_alloc(ix, &pl);
            if (rc == 0) rc = mcx_profile_attach(c, pl, max_dup, max_clip);
        }
        (void)hipMemGetInfo(&hbm_free, &hbm_all);
        const size_t took = free0 > hbm_free ? free0 - hbm_free : 0;
        if (cap) hbm_free = std::min(hbm_free, cap > took ? cap - took : 0);
        const bool fits = rc == 0 && hbm_free >= kFitMargin;
        if (fits) {
            f.max_batch_reads = o.max_batch_reads; f.hbm_free_bytes = (int64_t)hbm_free; f.hbm_taken_bytes = (int64_t)took; f.single_detail_set = with_profile && !c->later.have;
            if (fit) *fit = f;
            *out = c;
            if (planes) *planes = pl;
            return 0;
        }
        const std::string why = rc ? std::string(mcx_last_error()) : std::to_string((double)hbm_free / 1e9).substr(0, 5) + " GB of HBM would be left";
        (void)hipGetLastError(); // (an allocation that failed is no error of the run)
        if (pl) mcx_planes_free(pl);
        if (c) mcx_ctx_free(c);
        if (rc && rc != MCX_ERR_DEVICE) return rc; // (not a matter of room)
        if (ix->d_rank2) {
            if ((rc = mcx_index_trim(ix, 1))) return rc;
            f.pair_records_trimmed = 1;
            fprintf(stderr, "[mcx fit] %s with batches of %lld reads: the index gives its pair records back (the seeding walk takes one base per step)\n", why.c_str(), (long long)o.max_batch_reads);
            continue;
        }
        if (o.max_batch_reads > ((int64_t)1 << 17)) {
            o.max_batch_reads = std::max<int64_t>((int64_t)1 << 17, (o.max_batch_reads / 2 + 199) / 200 * 200); // (whole 200-read chunks: a caller that cuts its run into batches of this size keeps the chunk boundaries)
            f.batch_halvings++;
            fprintf(stderr, "[mcx fit] %s: batches of %lld reads instead\n", why.c_str(), (long long)o.max_batch_reads);
            continue;
        }
        return fail(MCX_ERR_DEVICE, "mcx_ctx_create_fit: the device has no room for this run even with batches of " + std::to_string((long long)o.max_batch_reads) + " reads and without the pair records (" + why + ")");
    }
}

// key buffers and radix-sort scratch for n keys (grown on demand: a round's keys of all shards can outnumber a batch)
static int sort_reserve(mcx_ctx *c, uint64_t n)
{
    if (n <= c->keys_cap) return 0;
    for (int k = 0; k < 2; k++) { if (c->d_keys[k]) (void)hipFree(c->d_keys[k]); c->d_keys[k] = nullptr; }
    if (c->d_sort_tmp) { (void)hipFree(c->d_sort_tmp); c->d_sort_tmp = nullptr; }
    c->keys_cap = 0;
    int rc;
    for (int k = 0; k < 2; k++) if ((rc = dmalloc(&c->d_keys[k], n))) return rc;
    hipcub::DoubleBuffer<uint64_t> dk(c->d_keys[0], c->d_keys[1]);
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, c->sort_tmp_bytes, dk, (int64_t)n, 0, 64));
    HIP_TRY(hipMalloc(&c->d_sort_tmp, c->sort_tmp_bytes + 256));
    c->keys_cap = n;
    return 0;
}

// What the host has to do about a queued batch's bookkeeping once it is through: overflow checks, the records into the archives.
static int profile_collect(mcx_ctx *c)
{
    auto &L = c->later;
    if (!L.pending) return 0;
    L.pending = false;
    HIP_TRY(hipEventSynchronize(L.done));
    const uint32_t n_sp = L.h_cnt[CNT_TASKS], n_ev = L.h_cnt[CNT_RESCUE];
    if (n_sp > c->sparse_cap || n_ev > L.ev_cap) return fail(MCX_ERR_CAPACITY, "profile: sparse record list overflow");
    if (L.h_cnt[CNT_UNSUP]) return fail(MCX_ERR_UNSUPPORTED, "an insertion or deletion of more than 255 bases in an alignment: its string does not fit a tally record");
    if (int rc = archive_append(c, c->arch, c->d_sparse, n_sp, L.stream)) return rc;
    if (int rc = archive_append(c, c->arch_ev, L.d_ev, n_ev, L.stream)) return rc;
    HIP_TRY(hipStreamSynchronize(L.stream)); // (the archives are read on the context's stream; the lists are the next batch's to fill)
    if (c->kn.timing) fprintf(stderr, "[mcx profile] %u tally records, %u events, %u listed fragments (queued behind the batch; looked at %.2f ms later)\n", n_sp, n_ev, L.h_cnt[CNT_RTASK],
                              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - L.t_queued).count());
    return 0;
}

// One shard: the whole bookkeeping of the batch just mapped — keys, sort, admission, accumulation — queued on the bookkeeping's stream; nothing of it is
// waited for here.  The number of keys stays on the device (k_prof_admit / k_prof_count take it from there), the break-point records' count stays where
// k_prof_keys left it.  The next batch's mapping writes the other set of detail records and flag bytes.
static int profile_queue(mcx_ctx *c)
{
    if (c->prof_settled) return fail(MCX_ERR_ARG, "the profile has been settled (mcx_profile_settle / _finalize): attach it again before mapping more reads");
    if (int rc = profile_collect(c)) return rc; // (the batch before this one: through long ago, it ran under this batch's kernels)
    auto &L = c->later;
    BatchRun &br = c->run;
    hipStream_t ps = L.stream;
    const IndexView &ix = c->idx->view;
    ProfView pv; pv.pl = planes_view(c->prof_planes, ix.G); pv.match = c->d_prof_match; pv.G = ix.G; pv.max_dup = c->prof_max_dup; pv.max_clip = c->prof_max_clip;
    SparseSink sink; sink.recs = c->d_sparse; sink.n = L.d_cnt + CNT_TASKS; sink.cap = c->sparse_cap; sink.refused = L.d_cnt + CNT_UNSUP;
    const uint32_t n = br.rb.n_reads;
    ReadBatch rb; rb.bases = L.d_keep_bases; rb.off = L.d_keep_off; rb.n_reads = n;
    HIP_TRY(hipEventSynchronize(L.kept)); // (the caller's buffer is the caller's again: the copy was made beside the batch's first kernels)
    HIP_TRY(hipEventRecord(L.go, c->stream));
    HIP_TRY(hipStreamWaitEvent(ps, L.go, 0));
    HIP_TRY(hipMemsetAsync(L.d_cnt, 0, CNT_N * sizeof(uint32_t), ps));
    k_prof_keys<<<(n + 255) / 256, 256, 0, ps>>>(c->d_detail, c->dlay, rb, ix, pv, sink, c->d_keys[0], L.d_cnt + CNT_OV);
    hipcub::DoubleBuffer<uint64_t> dk(c->d_keys[0], c->d_keys[1]);
    size_t tb = c->sort_tmp_bytes;
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(c->d_sort_tmp, tb, dk, (int64_t)n, 0, 64, ps));
    const uint64_t *d_keys = dk.Current(); // (the keys of the reads that do not reach the duplicate check are ~0 and sort behind the others)
    // the counters of the second half start at zero, but for the two the first half hands over: the keys' number, the break-point records'
    static_assert(CNT_OV < CNT_N && CNT_TASKS < CNT_N, "counter indices");
    k_zero_but<<<1, 64, 0, ps>>>(L.d_cnt, CNT_N, CNT_OV, CNT_TASKS);
    k_prof_admit<<<(n + 255) / 256, 256, 0, ps>>>(d_keys, 0, pv, c->d_admit, 0u, n, L.d_cnt + CNT_OV);
    k_prof_count<<<(n + 255) / 256, 256, 0, ps>>>(d_keys, 0, pv, L.d_cnt + CNT_OV);
    ColList cols; cols.items = c->d_prof_items; cols.n = L.d_cnt + CNT_RTASK; cols.cap = c->prof_items_cap;
    // (a smaller grid for the two — 512 to 4096 workgroups, the batch walked in strides — changes nothing: 28.8-29.3 ms per batch either way; what the bookkeeping
    //  and the mapping beside it share is the memory system's rate of scattered line transfers, not the CUs)
    k_prof_accum<<<(n + 255) / 256, 256, 0, ps>>>(c->d_detail, c->dlay, rb, ix, pv, sink, c->d_admit, br.paired, cols);
    k_prof_cols<<<4096, 256, 0, ps>>>(c->d_detail, c->dlay, rb, ix, pv, sink, cols);
    if (br.paired) k_prof_disc<<<(n / 2 + 255) / 256, 256, 0, ps>>>(c->d_detail, c->dlay, n / 2, br.read_base / 2, L.d_ev, L.d_cnt + CNT_RESCUE, L.ev_cap);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(L.h_cnt, L.d_cnt, CNT_N * sizeof(uint32_t), hipMemcpyDeviceToHost, ps));
    HIP_TRY(hipEventRecord(L.done, ps));
    L.pending = true; L.t_queued = std::chrono::steady_clock::now();
    std::swap(c->d_detail, L.d_detail_alt); std::swap(c->d_admit, L.d_admit_alt); // the next batch's mapping writes the other set,
    std::swap(L.d_keep_bases, L.d_keep_bases_alt); std::swap(L.d_keep_off, L.d_keep_off_alt); // its reads are kept in the other copy
    return 0;
}

// break points, clip gate, and the sorted (start position, read) keys of the reads that reach the duplicate check
static int profile_keys(mcx_ctx *c)
{
    BatchRun &br = c->run;
    hipStream_t s = c->stream;
    const IndexView &ix = c->idx->view;
    ProfView pv; pv.pl = planes_view(c->prof_planes, ix.G); pv.match = c->d_prof_match; pv.G = ix.G; pv.max_dup = c->prof_max_dup; pv.max_clip = c->prof_max_clip;
    SparseSink sink; sink.recs = c->d_sparse; sink.n = c->d_cnt + CNT_TASKS; sink.cap = c->sparse_cap; sink.refused = c->d_cnt + CNT_UNSUP;
    const uint32_t n = br.rb.n_reads;
    HIP_TRY(hipMemsetAsync(c->d_cnt, 0, CNT_N * sizeof(uint32_t), s));
    k_prof_keys<<<(n + 255) / 256, 256, 0, s>>>(c->d_detail, c->dlay, br.rb, ix, pv, sink, c->d_keys[0], c->d_cnt + CNT_OV);
    hipcub::DoubleBuffer<uint64_t> dk(c->d_keys[0], c->d_keys[1]);
    size_t tb = c->sort_tmp_bytes;
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(c->d_sort_tmp, tb, dk, (int64_t)n, 0, 64, s));
    HIP_TRY(hipMemcpyAsync(c->h_cnt, c->d_cnt, CNT_N * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    br.d_sorted_keys = dk.Current();
    br.n_keys = c->h_cnt[CNT_OV];       // the keys of the other reads are ~0 and sort behind them
    br.n_sparse_keys = c->h_cnt[CNT_TASKS]; // break-point records so far; the accumulation appends behind them
    return 0;
}

// the keys of a round this shard has no reads in: only the run's readCount moves
static int profile_foreign(mcx_ctx *c, const uint64_t *h_all, uint64_t n_all)
{
    if (n_all == 0) return 0;
    hipStream_t s = c->stream;
    ProfView pv; pv.pl = planes_view(c->prof_planes, c->idx->view.G); pv.match = c->d_prof_match; pv.G = c->idx->view.G; pv.max_dup = c->prof_max_dup; pv.max_clip = c->prof_max_clip;
    int rc = sort_reserve(c, n_all);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_keys[0], h_all, n_all * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    hipcub::DoubleBuffer<uint64_t> dk(c->d_keys[0], c->d_keys[1]);
    size_t tb = c->sort_tmp_bytes;
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(c->d_sort_tmp, tb, dk, (int64_t)n_all, 0, 64, s));
    k_prof_count<<<(unsigned)((n_all + 255) / 256), 256, 0, s>>>(dk.Current(), n_all, pv);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

// the archived records leave HBM: through the two halves of a page-locked bounce buffer (a copy from HBM straight into
// pageable memory runs at ~1 GB/s), the next piece in flight while this one is copied out
static int archive_flush(mcx_ctx *c, mcx_ctx::Archive &a)
{
    if (a.n == 0) return 0;
    hipStream_t s = c->stream;
    std::vector<mcx_sparse_rec> &host = *a.host;
    const bool tallies = &a == &c->arch;
    if (tallies) host.resize(c->n_tally); // (what the last mcx_profile_sparse* call appended for its caller goes again)
    host.reserve(host.size() + a.n); // (no resize: that would be one thread zeroing fresh pages before the copy touches them again)
    const uint64_t half = c->sparse_pin_recs / 2;
    const uint64_t n_piece = (a.n + half - 1) / half;
    auto start = [&](uint64_t k) -> hipError_t {
        const uint64_t lo = k * half, m = std::min<uint64_t>(half, a.n - lo);
        return hipMemcpyAsync(c->h_sparse_pin + (k & 1) * half, a.d + lo, m * sizeof(SparseRec), hipMemcpyDeviceToHost, s);
    };
    HIP_TRY(start(0));
    for (uint64_t k = 0; k < n_piece; k++) {
        HIP_TRY(hipStreamSynchronize(s));
        if (k + 1 < n_piece) HIP_TRY(start(k + 1));
        const uint64_t lo = k * half, m = std::min<uint64_t>(half, a.n - lo);
        const mcx_sparse_rec *piece = (const mcx_sparse_rec *)(c->h_sparse_pin + (k & 1) * half);
        host.insert(host.end(), piece, piece + m);
    }
    a.n = 0;
    if (tallies) c->n_tally = host.size();
    return 0;
}

static int sparse_flush(mcx_ctx *c)
{
    if (int rc = profile_collect(c)) return rc;
    c->h_sparse.resize(c->n_tally);
    if (int rc = archive_flush(c, c->arch)) return rc;
    return archive_flush(c, c->arch_ev);
}

// n records at d_src join an archive (on the stream); an archive that cannot grow any more goes to the host first
static int archive_append(mcx_ctx *c, mcx_ctx::Archive &a, const SparseRec *d_src, uint64_t n, hipStream_t on)
{
    if (n == 0) return 0;
    hipStream_t s = on ? on : c->stream;
    if (a.n + n > a.cap) {
        const uint64_t want = std::max<uint64_t>({2 * a.cap, a.n + n, &a == &c->arch ? (uint64_t)1 << 22 : (uint64_t)1 << 18});
        SparseRec *grown = nullptr;
        if (want <= c->arch_limit && hipMalloc((void **)&grown, want * sizeof(SparseRec)) == hipSuccess) {
            if (a.n) HIP_TRY(hipMemcpyAsync(grown, a.d, a.n * sizeof(SparseRec), hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (a.d) (void)hipFree(a.d);
            a.d = grown; a.cap = want;
        } else {
            (void)hipGetLastError();
            if (c->kn.timing) fprintf(stderr, "[mcx profile] no room in HBM for a larger record archive (%llu records wanted): what it holds goes to the host now, batch by batch from here on\n", (unsigned long long)want);
            int rc = archive_flush(c, a); // no room for a larger archive: what it holds goes to the host now
            if (rc) return rc;
            if (n > a.cap) {
                if (a.d) (void)hipFree(a.d);
                a.d = nullptr; a.cap = 0;
                HIP_TRY(hipMalloc((void **)&a.d, (size_t)n * sizeof(SparseRec)));
                a.cap = n;
            }
        }
    }
    HIP_TRY(hipMemcpyAsync(a.d + a.n, d_src, (size_t)n * sizeof(SparseRec), hipMemcpyDeviceToDevice, s));
    a.n += n;
    return 0;
}

// admission over `all` keys (null: the batch's own, already sorted on the device), then the accumulation of the own reads
static int profile_accumulate(mcx_ctx *c, const uint64_t *h_all, uint64_t n_all, uint32_t slot_stride, uint32_t own_slot)
{
    if (c->prof_settled) return fail(MCX_ERR_ARG, "the profile has been settled (mcx_profile_settle / _finalize): attach it again before mapping more reads");
    BatchRun &br = c->run;
    hipStream_t s = c->stream;
    const IndexView &ix = c->idx->view;
    ProfView pv; pv.pl = planes_view(c->prof_planes, ix.G); pv.match = c->d_prof_match; pv.G = ix.G; pv.max_dup = c->prof_max_dup; pv.max_clip = c->prof_max_clip;
    SparseSink sink; sink.recs = c->d_sparse; sink.n = c->d_cnt + CNT_TASKS; sink.cap = c->sparse_cap; sink.refused = c->d_cnt + CNT_UNSUP;
    const uint32_t n = br.rb.n_reads;
    const int paired = br.paired;
    const uint64_t *d_keys = br.d_sorted_keys;
    uint64_t nk = br.n_keys;
    uint32_t own_lo = 0;
    if (h_all) {
        int rc = sort_reserve(c, std::max<uint64_t>(n_all, 1)); // (invalidates br.d_sorted_keys: the own keys are part of h_all)
        if (rc) return rc;
        if (n_all) HIP_TRY(hipMemcpyAsync(c->d_keys[0], h_all, n_all * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        hipcub::DoubleBuffer<uint64_t> dk(c->d_keys[0], c->d_keys[1]);
        size_t tb = c->sort_tmp_bytes;
        if (n_all) HIP_TRY(hipcub::DeviceRadixSort::SortKeys(c->d_sort_tmp, tb, dk, (int64_t)n_all, 0, 64, s));
        d_keys = dk.Current(); nk = n_all; own_lo = own_slot * slot_stride;
    }
    HIP_TRY(hipMemsetAsync(c->d_cnt, 0, CNT_N * sizeof(uint32_t), s));
    HIP_TRY(hipMemcpyAsync(c->d_cnt + CNT_TASKS, &br.n_sparse_keys, sizeof(uint32_t), hipMemcpyHostToDevice, s)); // (pageable source: copied before the call returns)
    if (nk) {
        k_prof_admit<<<(unsigned)((nk + 255) / 256), 256, 0, s>>>(d_keys, nk, pv, c->d_admit, own_lo, n);
        k_prof_count<<<(unsigned)((nk + 255) / 256), 256, 0, s>>>(d_keys, nk, pv);
    }
    ColList cols; cols.items = c->d_prof_items; cols.n = c->d_cnt + CNT_RTASK; cols.cap = c->prof_items_cap;
    k_prof_accum<<<(n + 255) / 256, 256, 0, s>>>(c->d_detail, c->dlay, br.rb, ix, pv, sink, c->d_admit, paired, cols); // (bit 1 of a flag byte: k_pack_reads')
    k_prof_cols<<<4096, 256, 0, s>>>(c->d_detail, c->dlay, br.rb, ix, pv, sink, cols);
    SparseRec *d_ev = (SparseRec *)c->d_tasks; // the SA task list is idle now
    const uint32_t ev_cap = (uint32_t)std::min<uint64_t>((uint64_t)c->task_cap * sizeof(uint2) / sizeof(SparseRec), 0x7fffffffu);
    if (paired) k_prof_disc<<<(n / 2 + 255) / 256, 256, 0, s>>>(c->d_detail, c->dlay, n / 2, br.read_base / 2, d_ev, c->d_cnt + CNT_RESCUE, ev_cap);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_cnt, c->d_cnt, CNT_N * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint32_t n_sp = c->h_cnt[CNT_TASKS], n_ev = c->h_cnt[CNT_RESCUE];
    if (n_sp > c->sparse_cap || n_ev > ev_cap) return fail(MCX_ERR_CAPACITY, "profile: sparse record list overflow");
    if (c->h_cnt[CNT_UNSUP]) return fail(MCX_ERR_UNSUPPORTED, "an insertion or deletion of more than 255 bases in an alignment: its string does not fit a tally record");
    // the records stay in HBM until somebody asks for them (mcx_profile_sparse*)
    const auto t_app = std::chrono::steady_clock::now();
    if (int rc = archive_append(c, c->arch, c->d_sparse, n_sp)) return rc;
    const int rc_ev = archive_append(c, c->arch_ev, d_ev, n_ev);
    if (c->kn.timing) fprintf(stderr, "[mcx profile] %u tally records, %u events, %u listed fragments; archives %.2f ms\n", n_sp, n_ev, c->h_cnt[CNT_RTASK],
                                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_app).count());
    return rc_ev;
}

// the planes kept as differences become counts (mcx_profile.h): once per run, after its last batch
extern "C" int mcx_profile_settle(mcx_ctx *c)
{
    if (!c) return fail(MCX_ERR_ARG, "mcx_profile_settle: null argument");
    if (c->prof_broken) return fail(MCX_ERR_DEVICE, "an earlier mcx_profile_settle failed half way: the planes are neither differences nor counts; attach the profile again");
    if (!c->prof_planes || c->prof_settled) return 0;
    HIP_TRY(hipSetDevice(c->idx->device));
    if (int rc = profile_collect(c)) return rc; // (the last batch's bookkeeping)
    hipStream_t s = c->stream;
    const IndexView &ix = c->idx->view;
    const size_t G = (size_t)ix.G;
    const PlanesView pl = planes_view(c->prof_planes, ix.G);
    size_t tb = 0, tb16 = 0;
    HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, tb, pl.multi, pl.multi, G, s));
    HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, tb16, c->d_prof_match, c->d_prof_match, G, s));
    tb = std::max(tb, tb16);
    void *tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, tb + 256)); // (nothing touched yet: a retry is safe)
    hipError_t e = hipcub::DeviceScan::InclusiveSum(tmp, tb, pl.multi, pl.multi, G, s);
    // the 16-bit difference planes: their words back to two differences each (mcx_planes.h), then the scan modulo 2^16
    uint16_t *diffs[5] = {pl.h(kPlF1), pl.h(kPlR2), pl.h(kPlF2), pl.h(kPlR1), c->d_prof_match};
    for (int k = 0; k < 5 && e == hipSuccess; k++) {
        k_prof_decode<<<8192, 256, 0, s>>>((uint32_t *)diffs[k], pl.stride / 2);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipcub::DeviceScan::InclusiveSum(tmp, tb, diffs[k], diffs[k], G, s);
    }
    if (e == hipSuccess) {
        ProfView pv; pv.pl = pl; pv.match = c->d_prof_match; pv.G = ix.G; pv.max_dup = c->prof_max_dup; pv.max_clip = c->prof_max_clip;
        k_prof_fold<<<8192, 256, 0, s>>>(ix, pv);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    if (e != hipSuccess) { c->prof_broken = c->prof_settled = true; HIP_TRY(e); } // (a second settle must not scan the scanned planes again)
    (void)hipFree(c->d_prof_match); c->d_prof_match = nullptr; // (6 GB at 3.1 Gbp: the variant caller's scans want the room)
    c->prof_settled = true;
    return 0;
}

extern "C" int mcx_profile_finalize(mcx_ctx *c, uint32_t *d_planes)
{
    if (!c || !d_planes) return fail(MCX_ERR_ARG, "mcx_profile_finalize: null argument");
    HIP_TRY(hipSetDevice(c->idx->device));
    if (d_planes == c->prof_planes) { if (int rc = mcx_profile_settle(c)) return rc; }
    k_prof_finalize<<<dim3(4096, kPlanes), 256, 0, c->stream>>>(planes_view(d_planes, c->idx->view.G), c->prof_max_dup);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int mcx_profile_sparse(mcx_ctx *c, const mcx_sparse_rec **recs, uint64_t *n)
{
    if (!c || !recs || !n) return fail(MCX_ERR_ARG, "mcx_profile_sparse: null argument");
    HIP_TRY(hipSetDevice(c->idx->device));
    if (int rc = sparse_flush(c)) return rc;
    mcx_disc_resolve(c->h_events.data(), c->h_events.size(), c->idx->view.G, c->h_sparse); // (appended behind the tallies: no second copy of them)
    *recs = c->h_sparse.data(); *n = c->h_sparse.size();
    return 0;
}

extern "C" int mcx_profile_sparse_shard(mcx_ctx *c, const mcx_sparse_rec **recs, uint64_t *n)
{
    if (!c || !recs || !n) return fail(MCX_ERR_ARG, "mcx_profile_sparse_shard: null argument");
    HIP_TRY(hipSetDevice(c->idx->device));
    if (int rc = sparse_flush(c)) return rc;
    c->h_sparse.insert(c->h_sparse.end(), c->h_events.begin(), c->h_events.end());
    *recs = c->h_sparse.data(); *n = c->h_sparse.size();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// per-call drop-ins: BWT_Search and nw/ksw2 alignment as batches
// ---------------------------------------------------------------------------------------------
__global__ void k_bwt_search(IndexView ix, const uint8_t *seqs, const uint32_t *off, const int32_t *start, uint32_t n,
                             int32_t *len, int32_t *freq, uint64_t *loc)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *seq = seqs + off[i];
    const int stop = (int)(off[i + 1] - off[i]);
    // BWT_Search (bwt_search.cpp:121-164), one call
    int p0 = start[i], c0 = seq[p0];
    uint64_t x0 = ix.L2[c0] + 1, x1 = ix.L2[3 - c0] + 1, x2 = ix.L2[c0 + 1] - ix.L2[c0];
    int pos;
    for (pos = p0 + 1; pos < stop; pos++) {
        int c = seq[pos];
        if (c > 3) break;
        uint64_t tk[4], tl[4]; int nb;
        fm_2occ4(ix, x1 - 1, x1 - 1 + x2, tk, tl, nb);
        int b = 3 - c;
        uint64_t n2 = tl[b] - tk[b];
        if (n2 == 0) break;
        uint64_t n0 = x0 + ((x1 <= ix.primary && x1 + x2 - 1 >= ix.primary) ? 1 : 0);
        for (int bb = 3; bb > b; bb--) n0 += tl[bb] - tk[bb];
        x0 = n0; x1 = ix.L2[b] + 1 + tk[b]; x2 = n2;
    }
    int l = pos - p0, f = 0;
    if (l >= kMinSeedLength && x2 <= (uint64_t)kOccThr) f = (int)x2;
    len[i] = l; freq[i] = f;
    for (int k = 0; k < f; k++) { int lf = 0; loc[(uint64_t)i * kOccThr + k] = fm_sa(ix, x0 + k, lf); }
}

extern "C" int mcx_bwt_search_batch(mcx_ctx *c, const uint8_t *seqs, const uint32_t *seq_off, const int32_t *start,
                                    uint32_t n, int32_t *len, int32_t *freq, uint64_t *loc)
{
    if (!c || !seqs || !seq_off || !start || !len || !freq || !loc) return fail(MCX_ERR_ARG, "mcx_bwt_search_batch: null argument");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->idx->device));
    uint8_t *d_seq = nullptr; uint32_t *d_off = nullptr; int32_t *d_start = nullptr, *d_len = nullptr, *d_freq = nullptr; uint64_t *d_loc = nullptr;
    int rc = 0;
    size_t nb = seq_off[n];
    if ((rc = dmalloc(&d_seq, nb + 16)) || (rc = dmalloc(&d_off, (size_t)n + 1)) || (rc = dmalloc(&d_start, n)) ||
        (rc = dmalloc(&d_len, n)) || (rc = dmalloc(&d_freq, n)) || (rc = dmalloc(&d_loc, (size_t)n * kOccThr))) return rc;
    HIP_TRY(hipMemcpy(d_seq, seqs, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_off, seq_off, ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_start, start, (size_t)n * 4, hipMemcpyHostToDevice));
    k_bwt_search<<<(n + 255) / 256, 256, 0, c->stream>>>(c->idx->view, d_seq, d_off, d_start, n, d_len, d_freq, d_loc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(len, d_len, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(freq, d_freq, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(loc, d_loc, (size_t)n * kOccThr * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_seq); (void)hipFree(d_off); (void)hipFree(d_start); (void)hipFree(d_len); (void)hipFree(d_freq); (void)hipFree(d_loc);
    return 0;
}

// stand-alone extension (nw_alignment / ksw2_alignment as a batch): strings come from user buffers
struct ExtArgs {
    const uint8_t *q, *t;
    const uint32_t *q_off, *t_off;
    uint8_t *ops; int32_t *ops_len, *score;
    uint32_t n;
    int use_nw;
};

template <int K>
__global__ void __launch_bounds__(64) k_extend(ExtArgs a, uint8_t *scratch, uint64_t stride, int t_lo, int t_hi)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kDpLdsSeq + kDpLdsDir];
    uint8_t *spill = scratch + (uint64_t)blockIdx.x * stride;
    const int lane = threadIdx.x;
    for (uint32_t jb = blockIdx.x; jb < a.n; jb += gridDim.x) {
        const int m = (int)(a.q_off[jb + 1] - a.q_off[jb]), n = (int)(a.t_off[jb + 1] - a.t_off[jb]);
        if (n <= t_lo || n > t_hi || m <= 0) continue;
        const DpBuf b = dp_buffers(m, n, lds, spill);
        for (int i = lane; i < m; i += 64) b.q[i] = (uint8_t)nt4_code(a.q[a.q_off[jb] + i]);
        for (int i = lane; i < n; i += 64) b.t[i] = (uint8_t)nt4_code(a.t[a.t_off[jb] + i]);
        __syncthreads();
        uint8_t *dst = a.ops + a.q_off[jb] + a.t_off[jb];
        int score = 0;
        const int w = dp_core<K, 64, true>(a.use_nw != 0, m, n, b, dst, &score, nullptr, 0u);
        const int L = m + n - w;
        for (int base = 0; base < L; base += 64) { // move the string to the front of its area
            uint8_t v = base + lane < L ? dst[w + base + lane] : 0;
            __syncthreads();
            if (base + lane < L) dst[base + lane] = v;
            __syncthreads();
        }
        if (lane == 0) { a.ops_len[jb] = L; a.score[jb] = score; }
    }
}

extern "C" int mcx_extend_batch(mcx_ctx *c, int alg, const uint8_t *q, const uint32_t *q_off, const uint8_t *t,
                                const uint32_t *t_off, uint32_t n, uint8_t *ops, int32_t *ops_len, int32_t *score)
{
    if (!c || !q || !q_off || !t || !t_off || !ops || !ops_len || !score) return fail(MCX_ERR_ARG, "mcx_extend_batch: null argument");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->idx->device));
    for (uint32_t i = 0; i < n; i++) {
        if (t_off[i + 1] - t_off[i] > 1024 || q_off[i + 1] - q_off[i] > 2048 || t_off[i + 1] == t_off[i] || q_off[i + 1] == q_off[i])
            return fail(MCX_ERR_UNSUPPORTED, "mcx_extend_batch: fragments must be 1..2048 (read) x 1..1024 (genome)");
    }
    ExtArgs a; a.n = n; a.use_nw = alg == 0;
    uint8_t *d_q = nullptr, *d_t = nullptr, *d_ops = nullptr; uint32_t *d_qo = nullptr, *d_to = nullptr; int32_t *d_len = nullptr, *d_sc = nullptr;
    const size_t nq = q_off[n], nt = t_off[n];
    int rc = 0;
    if ((rc = dmalloc(&d_q, nq + 16)) || (rc = dmalloc(&d_t, nt + 16)) || (rc = dmalloc(&d_ops, nq + nt + 16)) ||
        (rc = dmalloc(&d_qo, (size_t)n + 1)) || (rc = dmalloc(&d_to, (size_t)n + 1)) || (rc = dmalloc(&d_len, n)) || (rc = dmalloc(&d_sc, n))) return rc;
    HIP_TRY(hipMemcpy(d_q, q, nq, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_t, t, nt, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_qo, q_off, ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_to, t_off, ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
    a.q = d_q; a.t = d_t; a.q_off = d_qo; a.t_off = d_to; a.ops = d_ops; a.ops_len = d_len; a.score = d_sc;
    k_extend<1><<<c->dp_blocks[0], 64, 0, c->stream>>>(a, c->d_dp_scratch[0], c->dp_stride[0], 0, 64);
    k_extend<4><<<c->dp_blocks[1], 64, 0, c->stream>>>(a, c->d_dp_scratch[1], c->dp_stride[1], 64, 256);
    k_extend<16><<<c->dp_blocks[2], 64, 0, c->stream>>>(a, c->d_dp_scratch[2], c->dp_stride[2], 256, 1024);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(ops, d_ops, nq + nt, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ops_len, d_len, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(score, d_sc, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_q); (void)hipFree(d_t); (void)hipFree(d_ops); (void)hipFree(d_qo); (void)hipFree(d_to); (void)hipFree(d_len); (void)hipFree(d_sc);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// files in, SAM out: MapCaller -i <prefix> -f A [-f2 B] -sam out  (main.cpp:212-321, Mapping() ReadMapping.cpp:689-747)
// ---------------------------------------------------------------------------------------------
extern "C" int mcx_cigar_words(mcx_ctx *c, uint32_t *n_words)
{
    if (!c || !n_words) return fail(MCX_ERR_ARG, "mcx_cigar_words: null argument");
    *n_words = c->run.cig_words;
    return 0;
}

const mcx_index *mcx_ctx_index(const mcx_ctx *c) { return c->idx; }
int mcx_ctx_max_read_len(const mcx_ctx *c) { return c->rlen_max; }
bool mcx_ctx_has_profile(const mcx_ctx *c) { return c->prof_planes != nullptr; }
uint64_t mcx_ctx_max_reads(const mcx_ctx *c) { return c->max_reads; }
void **mcx_ctx_files_slot(mcx_ctx *c, void (*drop)(void *)) { c->files_drop = drop; return &c->files_state; }
void *mcx_pinned_alloc(size_t bytes) { void *p = nullptr; return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr; }
void mcx_pinned_free(void *p) { if (p) (void)hipHostFree(p); }
