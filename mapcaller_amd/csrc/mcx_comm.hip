// mapcaller_amd/csrc/mcx_comm.hip — libmcx_comm.so: the profile reduce of a multi-GPU run over RCCL.
//
// The reference keeps one MappingRecordArr in host memory that every mapping thread updates under
// ProfileLock (src/AlignmentProfile.cpp:41-242, src/ReadMapping.cpp:562-573).  Here every GPU keeps its
// own counter planes while it maps and they are summed once, onto the GPU that calls the variants:
// ncclReduce over xGMI of the planes as they lie in HBM (mcx_planes.h: multi_hit in 32 bits, the others in 16 — RCCL has no
// 16-bit integer sum, so those travel as words of two positions; 20 bytes per position on the wire; the readCount plane is
// already the run's on every rank).  xGMI is point to point, so the reduce is issued in pieces of at most 2^28 words
// (1 GiB): large enough to run at link speed, small enough for a 32-bit count and to let the rings of consecutive
// pieces overlap.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mcx_comm.h"
#include "mcx_build.h"
#include "mcx_planes.h"

namespace {

struct Group { // ranks that share a process without RCCL (two ranks on one device)
    std::mutex m; std::condition_variable cv;
    int size = 0, arrived = 0, refs = 0;
    uint64_t gen = 0;
    std::atomic<bool> failed{false}; // a rank of the group failed inside a collective: the others give up after the barrier
    std::vector<void *> ptr;
    void barrier()
    {
        std::unique_lock<std::mutex> l(m);
        const uint64_t g = gen;
        if (++arrived == size) { arrived = 0; gen++; cv.notify_all(); }
        else cv.wait(l, [&] { return gen != g; });
    }
};

__global__ void k_add_planes(uint32_t *dst, const uint32_t *src, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

// the same for words that hold two 16-bit counters (mcx_planes.h): each half by itself, modulo 2^16
__global__ void k_add_halves(uint32_t *dst, const uint32_t *src, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t a = dst[i], b = src[i];
        dst[i] = ((a + b) & 0xFFFFu) | ((a & 0xFFFF0000u) + (b & 0xFFFF0000u));
    }
}

// A C G T before they are summed as words: every rank's counters clamped to 4095 — sixteen ranks cannot carry into the
// neighbouring half, and min(sum of min(x, 4095), 4095) is min(sum of x, 4095), which is what mcx_profile_finalize leaves.
__global__ void k_clamp_halves(uint32_t *w, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = w[i], lo = v & 0xFFFFu, hi = v >> 16;
        if (lo > 4095u || hi > 4095u) w[i] = (lo < 4095u ? lo : 4095u) | ((hi < 4095u ? hi : 4095u) << 16);
    }
}

// the largest 16-bit counter among n words: the strand counters wrap at 2^16, so their words can be summed as words only when no
// low half can carry (this maximum over all ranks, times the number of ranks, stays below 2^16)
__global__ void k_half_max(const uint32_t *w, uint64_t n, uint32_t *out)
{
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = w[i], x = v & 0xFFFFu, y = v >> 16;
        m = m > x ? m : x; m = m > y ? m : y;
    }
    for (int o = 32; o > 0; o >>= 1) { const uint32_t other = (uint32_t)__shfl_down((int)m, o, 64); m = m > other ? m : other; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// the scattered reduce: the slices of this rank's share that the peers sent (slice j of `in` at j * stride), summed word by word
__global__ void k_sum_slices(const uint32_t *in, int n_slices, uint64_t stride, uint64_t n, uint32_t *out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = 0;
        for (int j = 0; j < n_slices; j++) v += in[(uint64_t)j * stride + i];
        out[i] = v;
    }
}

// when they can carry: a piece of the words widened to one counter per u32, summed, and narrowed again on the root
__global__ void k_widen(const uint32_t *w, uint64_t n, uint32_t *out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { const uint32_t v = w[i]; out[2 * i] = v & 0xFFFFu; out[2 * i + 1] = v >> 16; }
}
__global__ void k_narrow(const uint32_t *in, uint64_t n, uint32_t *w)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) w[i] = (in[2 * i] & 0xFFFFu) | (in[2 * i + 1] << 16);
}

int fail_hip(const char *what, hipError_t e) { return mcx_set_error(MCX_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); }
int fail_nccl(const char *what, ncclResult_t e) { return mcx_set_error(MCX_ERR_DEVICE, std::string(what) + ": " + ncclGetErrorString(e)); }

} // namespace

struct mcx_comm {
    ncclComm_t nccl = nullptr;
    Group *group = nullptr;
    int rank = 0, size = 1, device = 0;
    hipStream_t stream = nullptr;
};

extern "C" int32_t mcx_comm_rank(const mcx_comm *c) { return c ? c->rank : 0; }
extern "C" int32_t mcx_comm_size(const mcx_comm *c) { return c ? c->size : 1; }

extern "C" int mcx_comm_init_all(int32_t n, const int32_t *devices, mcx_comm **out)
{
    if (n < 1 || !devices || !out) return mcx_set_error(MCX_ERR_ARG, "mcx_comm_init_all: bad argument");
    bool distinct = true;
    for (int a = 0; a < n; a++) for (int b = a + 1; b < n; b++) if (devices[a] == devices[b]) distinct = false;
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    Group *g = nullptr;
    if (distinct) {
        std::vector<int> devs(devices, devices + n);
        ncclResult_t e = ncclCommInitAll(comms.data(), n, devs.data());
        if (e != ncclSuccess) return fail_nccl("ncclCommInitAll", e);
    } else {
        g = new Group();
        g->size = n; g->refs = n; g->ptr.assign((size_t)n, nullptr);
    }
    for (int r = 0; r < n; r++) {
        mcx_comm *c = new mcx_comm();
        c->nccl = comms[(size_t)r]; c->group = g; c->rank = r; c->size = n; c->device = devices[r];
        out[r] = c;
    }
    return 0;
}

extern "C" int mcx_comm_unique_id(uint8_t id[MCX_COMM_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) <= MCX_COMM_ID_BYTES, "ncclUniqueId must fit MCX_COMM_ID_BYTES");
    if (!id) return mcx_set_error(MCX_ERR_ARG, "mcx_comm_unique_id: null argument");
    ncclUniqueId u;
    ncclResult_t e = ncclGetUniqueId(&u);
    if (e != ncclSuccess) return fail_nccl("ncclGetUniqueId", e);
    memset(id, 0, MCX_COMM_ID_BYTES);
    memcpy(id, &u, sizeof u);
    return 0;
}

extern "C" int mcx_comm_init_rank(const uint8_t id[MCX_COMM_ID_BYTES], int32_t rank, int32_t size, int32_t device, mcx_comm **out)
{
    if (!id || !out || size < 1 || rank < 0 || rank >= size) return mcx_set_error(MCX_ERR_ARG, "mcx_comm_init_rank: bad argument");
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return fail_hip("hipSetDevice", he);
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t nc = nullptr;
    ncclResult_t e = ncclCommInitRank(&nc, size, u, rank);
    if (e != ncclSuccess) return fail_nccl("ncclCommInitRank", e);
    mcx_comm *c = new mcx_comm();
    c->nccl = nc; c->rank = rank; c->size = size; c->device = device;
    *out = c;
    return 0;
}

extern "C" void mcx_comm_free(mcx_comm *c)
{
    if (!c) return;
    if (c->stream) { (void)hipSetDevice(c->device); (void)hipStreamDestroy(c->stream); }
    if (c->nccl) (void)ncclCommDestroy(c->nccl);
    if (c->group) {
        bool last;
        { std::unique_lock<std::mutex> l(c->group->m); last = --c->group->refs == 0; }
        if (last) delete c->group;
    }
    delete c;
}

static int comm_stream(mcx_comm *c)
{
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail_hip("hipSetDevice", e);
    if (!c->stream && (e = hipStreamCreate(&c->stream)) != hipSuccess) return fail_hip("hipStreamCreate", e);
    return 0;
}

extern "C" int mcx_profile_reduce(mcx_comm *c, uint32_t *d_planes, int64_t G, int32_t root, double *seconds)
{
    if (!c || !d_planes || G <= 0 || root < 0 || root >= c->size) return mcx_set_error(MCX_ERR_ARG, "mcx_profile_reduce: bad argument");
    const auto t0 = std::chrono::steady_clock::now();
    int rc = comm_stream(c);
    if (rc) return rc;
    // the planes as words (mcx_planes.h): multi_hit one counter a word; A C G T | readCount | F1 R2 F2 R1 two positions a word
    const mcx::PlanesView pl = mcx::planes_view(d_planes, G);
    const uint64_t hw = pl.stride / 2; // words per 16-bit plane
    uint32_t *acgt = (uint32_t *)pl.h(mcx::kPlA), *strands = (uint32_t *)pl.h(mcx::kPlF1);
    if (c->nccl) { // (one rank too: the same kernels and collectives — that is what a one-GPU box can test)
        if (c->size > 16) return mcx_set_error(MCX_ERR_UNSUPPORTED, "mcx_profile_reduce: more than 16 ranks (A C G T travel as 16-bit halves clamped to 4095)");
        const uint64_t piece = 1ull << 28;
        auto dead = [&](const char *what, ncclResult_t e) { // the peers sit in the collectives queued so far: abort the communicator so that they come back with an error
            const int r = fail_nccl(what, e);
            (void)ncclCommAbort(c->nccl); c->nccl = nullptr;
            return r;
        };
        auto dead_hip = [&](const char *what, hipError_t e) { const int r = fail_hip(what, e); (void)ncclCommAbort(c->nccl); c->nccl = nullptr; return r; };
        // Two ways for a piece of words onto the root.  One ncclReduce: a ring or tree of RCCL's choosing, every byte of the piece crosses ONE
        // inbound link of the root.  Scattered (three ranks and more; MCX_REDUCE_SCATTER=1 / 0 decides otherwise): the piece cut into one slice
        // per rank, the slices exchanged all to all — xGMI is a full mesh of point-to-point links, so all of a GPU's links carry a slice at once —,
        // summed by their owners, the sums sent to the root over all of its links: 2 x piece / ranks per link instead of the piece.
        const char *sc_env = getenv("MCX_REDUCE_SCATTER");
        const bool scatter = sc_env ? atoi(sc_env) != 0 : c->size >= 3;
        uint32_t *tmp = nullptr;
        const uint64_t per_max = (piece + (uint64_t)c->size - 1) / (uint64_t)c->size;
        if (scatter) {
            hipError_t he = hipMalloc((void **)&tmp, (size_t)((uint64_t)(c->size + 1) * per_max) * sizeof(uint32_t));
            if (he != hipSuccess) return dead_hip("mcx_profile_reduce (room for the scattered slices)", he);
        }
        struct Free { uint32_t *&p; ~Free() { if (p) (void)hipFree(p); } } free_tmp{tmp};
        auto reduce_words = [&](uint32_t *p, uint64_t n) -> int {
            for (uint64_t lo = 0; lo < n; lo += piece) {
                const uint64_t cnt = std::min<uint64_t>(piece, n - lo);
                if (!scatter) {
                    ncclResult_t e = ncclReduce(p + lo, p + lo, (size_t)cnt, ncclUint32, ncclSum, root, c->nccl, c->stream);
                    if (e != ncclSuccess) return dead("ncclReduce", e);
                    continue;
                }
                const uint64_t per = (cnt + (uint64_t)c->size - 1) / (uint64_t)c->size;
                auto len = [&](int j) { const uint64_t o = (uint64_t)j * per; return o < cnt ? std::min<uint64_t>(per, cnt - o) : 0ull; };
                uint32_t *own = tmp + (uint64_t)c->size * per_max;
                const uint64_t mine = len(c->rank);
                ncclResult_t e = ncclGroupStart();
                for (int j = 0; j < c->size && e == ncclSuccess; j++) {
                    if (j == c->rank) continue;
                    if (len(j)) e = ncclSend(p + lo + (uint64_t)j * per, (size_t)len(j), ncclUint32, j, c->nccl, c->stream);
                    if (e == ncclSuccess && mine) e = ncclRecv(tmp + (uint64_t)j * per_max, (size_t)mine, ncclUint32, j, c->nccl, c->stream);
                }
                if (e == ncclSuccess) e = ncclGroupEnd(); else (void)ncclGroupEnd();
                if (e != ncclSuccess) return dead("ncclSend / ncclRecv (slices)", e);
                if (mine) {
                    hipError_t he = hipMemcpyAsync(tmp + (uint64_t)c->rank * per_max, p + lo + (uint64_t)c->rank * per, (size_t)mine * 4, hipMemcpyDeviceToDevice, c->stream);
                    if (he != hipSuccess) return dead_hip("hipMemcpyAsync", he);
                    k_sum_slices<<<4096, 256, 0, c->stream>>>(tmp, c->size, per_max, mine, own);
                }
                e = ncclGroupStart();
                if (c->rank == root) {
                    for (int j = 0; j < c->size && e == ncclSuccess; j++) if (j != root && len(j)) e = ncclRecv(p + lo + (uint64_t)j * per, (size_t)len(j), ncclUint32, j, c->nccl, c->stream);
                } else if (mine) e = ncclSend(own, (size_t)mine, ncclUint32, root, c->nccl, c->stream);
                if (e == ncclSuccess) e = ncclGroupEnd(); else (void)ncclGroupEnd();
                if (e != ncclSuccess) return dead("ncclSend / ncclRecv (sums)", e);
                if (c->rank == root && mine) {
                    hipError_t he = hipMemcpyAsync(p + lo + (uint64_t)root * per, own, (size_t)mine * 4, hipMemcpyDeviceToDevice, c->stream);
                    if (he != hipSuccess) return dead_hip("hipMemcpyAsync", he);
                }
            }
            return 0;
        };
        k_clamp_halves<<<8192, 256, 0, c->stream>>>(acgt, 4 * hw);
        if ((rc = reduce_words(acgt, 4 * hw))) return rc;
        // can the strand counters travel as words?  the largest one over all ranks decides, the same way on every rank
        bool strands_share = false;
        {
            uint32_t *d_top = nullptr, top = 0;
            hipError_t he = hipMalloc((void **)&d_top, sizeof(uint32_t));
            if (he == hipSuccess) he = hipMemsetAsync(d_top, 0, sizeof(uint32_t), c->stream);
            if (he != hipSuccess) return dead_hip("mcx_profile_reduce", he);
            k_half_max<<<4096, 256, 0, c->stream>>>(strands, 4 * hw, d_top);
            ncclResult_t e = ncclAllReduce(d_top, d_top, 1, ncclUint32, ncclMax, c->nccl, c->stream);
            if (e != ncclSuccess) { (void)hipFree(d_top); return dead("ncclAllReduce", e); }
            he = hipMemcpyAsync(&top, d_top, sizeof top, hipMemcpyDeviceToHost, c->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
            (void)hipFree(d_top);
            if (he != hipSuccess) return dead_hip("mcx_profile_reduce", he);
            strands_share = (uint64_t)top * (uint64_t)c->size <= 0xFFFFu && !getenv("MCX_REDUCE_WIDE"); // (MCX_REDUCE_WIDE: tests — the other way on a box with one GPU)
        }
        if (strands_share) { if ((rc = reduce_words(strands, 4 * hw))) return rc; }
        else { // (sums of independent deep runs: a counter per word on the wire, piece by piece)
            uint32_t *wide = nullptr;
            const uint64_t pw = 1ull << 27; // words per piece: 2^28 counters
            hipError_t he = hipMalloc((void **)&wide, (size_t)(2 * std::min<uint64_t>(pw, 4 * hw)) * sizeof(uint32_t));
            if (he != hipSuccess) return dead_hip("mcx_profile_reduce (room to widen the strand planes)", he);
            for (uint64_t lo = 0; lo < 4 * hw && rc == 0; lo += pw) {
                const uint64_t cnt = std::min<uint64_t>(pw, 4 * hw - lo);
                k_widen<<<8192, 256, 0, c->stream>>>(strands + lo, cnt, wide);
                ncclResult_t e = ncclReduce(wide, wide, (size_t)(2 * cnt), ncclUint32, ncclSum, root, c->nccl, c->stream);
                if (e != ncclSuccess) { rc = dead("ncclReduce", e); break; }
                if (c->rank == root) k_narrow<<<8192, 256, 0, c->stream>>>(wide, cnt, strands + lo);
            }
            if (rc == 0) (void)hipStreamSynchronize(c->stream);
            (void)hipFree(wide);
            if (rc) return rc;
        }
        if ((rc = reduce_words(pl.multi, pl.stride))) return rc;
        hipError_t he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) return dead_hip("hipStreamSynchronize", he);
        ncclResult_t ae = ncclSuccess;
        if (ncclCommGetAsyncError(c->nccl, &ae) == ncclSuccess && ae != ncclSuccess) { rc = fail_nccl("ncclReduce (asynchronous)", ae); (void)ncclCommAbort(c->nccl); c->nccl = nullptr; return rc; }
    } else if (c->size > 1) { // ranks of one process on a shared device: the root adds the others' planes itself
        Group &g = *c->group;
        g.ptr[(size_t)c->rank] = d_planes;
        // (every rank reaches both barriers whatever happens to it: a rank that returned early would leave the others waiting for ever)
        hipError_t he = hipDeviceSynchronize();
        if (he != hipSuccess) { rc = fail_hip("hipDeviceSynchronize", he); g.failed.store(true); }
        g.barrier();
        if (c->rank == root && !g.failed.load()) {
            for (int r = 0; r < c->size && rc == 0; r++) {
                if (r == root) continue;
                const mcx::PlanesView o = mcx::planes_view(g.ptr[(size_t)r], G);
                k_add_planes<<<4096, 256, 0, c->stream>>>(pl.multi, o.multi, pl.stride);
                k_add_halves<<<4096, 256, 0, c->stream>>>(acgt, (const uint32_t *)o.h(mcx::kPlA), 4 * hw);       // (the readCount plane between them is left alone)
                k_add_halves<<<4096, 256, 0, c->stream>>>(strands, (const uint32_t *)o.h(mcx::kPlF1), 4 * hw);
                if ((he = hipGetLastError()) != hipSuccess) rc = fail_hip("k_add_planes", he);
            }
            he = hipStreamSynchronize(c->stream);
            if (he != hipSuccess && rc == 0) rc = fail_hip("k_add_planes", he);
            if (rc) g.failed.store(true);
        }
        g.barrier();
        if (rc == 0 && g.failed.load()) rc = mcx_set_error(MCX_ERR_DEVICE, "mcx_profile_reduce: another rank of the group failed");
    }
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// ---- an mcx_exchange over the communicator -------------------------------------------------------------
namespace {
struct CommLink { mcx_comm *c; void *d_send = nullptr, *d_recv = nullptr; uint64_t cap = 0; };

int comm_allgather(void *user, const void *send, void *recv, uint64_t bytes)
{
    CommLink *l = (CommLink *)user;
    mcx_comm *c = l->c;
    if (comm_stream(c)) return MCX_ERR_DEVICE;
    if (bytes == 0) return 0;
    if (c->size == 1 || !c->nccl) {
        if (c->size == 1) { memcpy(recv, send, bytes); return 0; }
        return MCX_ERR_UNSUPPORTED; // ranks sharing a device use mcx_exchange_local
    }
    if (bytes > l->cap) {
        if (l->d_send) (void)hipFree(l->d_send);
        if (l->d_recv) (void)hipFree(l->d_recv);
        l->d_send = l->d_recv = nullptr; l->cap = 0;
        const uint64_t want = bytes + bytes / 2 + 4096;
        if (hipMalloc(&l->d_send, want) != hipSuccess || hipMalloc(&l->d_recv, want * (uint64_t)c->size) != hipSuccess) return MCX_ERR_DEVICE;
        l->cap = want;
    }
    if (hipMemcpyAsync(l->d_send, send, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return MCX_ERR_DEVICE;
    if (ncclAllGather(l->d_send, l->d_recv, (size_t)bytes, ncclUint8, c->nccl, c->stream) != ncclSuccess) return MCX_ERR_DEVICE;
    if (hipMemcpyAsync(recv, l->d_recv, bytes * (uint64_t)c->size, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return MCX_ERR_DEVICE;
    return hipStreamSynchronize(c->stream) == hipSuccess ? 0 : MCX_ERR_DEVICE;
}
} // namespace

extern "C" int mcx_comm_exchange(mcx_comm *c, mcx_exchange *out)
{
    if (!c || !out) return mcx_set_error(MCX_ERR_ARG, "mcx_comm_exchange: null argument");
    CommLink *l = new CommLink();
    l->c = c;
    out->user = l; out->rank = c->rank; out->size = c->size; out->allgather = comm_allgather;
    return 0;
}

extern "C" void mcx_comm_exchange_free(mcx_exchange *x)
{
    if (!x || !x->user) return;
    CommLink *l = (CommLink *)x->user;
    if (l->d_send) (void)hipFree(l->d_send);
    if (l->d_recv) (void)hipFree(l->d_recv);
    delete l;
    x->user = nullptr;
}
