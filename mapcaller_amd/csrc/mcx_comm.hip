// mapcaller_amd/csrc/mcx_comm.hip — libmcx_comm.so: the profile reduce of a multi-GPU run over RCCL.
//
// The reference keeps one MappingRecordArr in host memory that every mapping thread updates under
// ProfileLock (src/AlignmentProfile.cpp:41-242, src/ReadMapping.cpp:562-573).  Here every GPU keeps its
// own counter planes while it maps and they are summed once, onto the GPU that calls the variants:
// ncclReduce over xGMI, nine planes of u32 (the readCount plane is already the run's on every rank).
// xGMI is point to point, so the reduce is issued plane by plane in pieces of at most 2^28 elements
// (1 GiB): large enough to run at link speed, small enough for a 32-bit count and to let the
// rings of consecutive pieces overlap.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mcx_comm.h"
#include "mcx_build.h"

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

// Two 16-bit counters share a u32 on the wire (the planes are 12- and 16-bit fields once finalised, mcx_profile_finalize).
// clamp != 0: saturating counters (A C G T), every rank's value clamped to 4095 first — sixteen ranks cannot carry into the
// high half, and min(sum of min(x, 4095), 4095) is min(sum of x, 4095).  clamp == 0: the strand counters, which wrap at 2^16;
// used only when no low half can carry (k_low_max over all ranks, times the number of ranks, stays below 2^16).
__global__ void k_pack_planes(uint32_t *lo, const uint32_t *hi, uint64_t n, int clamp)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t a = lo[i], b = hi[i];
        if (clamp) { a = a < 4095u ? a : 4095u; b = b < 4095u ? b : 4095u; }
        lo[i] = (a & 0xFFFFu) | (b << 16);
    }
}

__global__ void k_unpack_planes(uint32_t *lo, uint32_t *hi, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t p = lo[i];
        lo[i] = p & 0xFFFFu; hi[i] = p >> 16;
    }
}

__global__ void k_low_max(const uint32_t *a, const uint32_t *b, uint64_t n, uint32_t *out)
{
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t x = a[i] & 0xFFFFu, y = b[i] & 0xFFFFu;
        m = m > x ? m : x; m = m > y ? m : y;
    }
    for (int o = 32; o > 0; o >>= 1) { const uint32_t other = (uint32_t)__shfl_down((int)m, o, 64); m = m > other ? m : other; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
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
    const int kPlanes = 10, kReadCount = 5;
    if (c->nccl) { // (one rank too: the same packing, collectives and unpacking — that is what a one-GPU box can test)
        const uint64_t piece = 1ull << 28, n = (uint64_t)G;
        auto plane = [&](int k) { return d_planes + (uint64_t)k * n; };
        auto dead = [&](const char *what, ncclResult_t e) { // the peers sit in the collectives queued so far: abort the communicator so that they come back with an error
            const int r = fail_nccl(what, e);
            (void)ncclCommAbort(c->nccl); c->nccl = nullptr;
            return r;
        };
        auto reduce_plane = [&](int k) -> int {
            uint32_t *p = plane(k);
            for (uint64_t lo = 0; lo < n; lo += piece) {
                const uint64_t cnt = std::min<uint64_t>(piece, n - lo);
                ncclResult_t e = ncclReduce(p + lo, p + lo, (size_t)cnt, ncclUint32, ncclSum, root, c->nccl, c->stream);
                if (e != ncclSuccess) return dead("ncclReduce", e);
            }
            return 0;
        };
        enum { pA = 0, pC, pG, pT, pMulti, pRC, pF1, pR2, pF2, pR1 };
        // can the strand planes share words?  the largest low half over all ranks decides, the same way on every rank
        bool strands_share = false;
        {
            uint32_t *d_top = nullptr, top = 0;
            hipError_t he = hipMalloc((void **)&d_top, sizeof(uint32_t));
            if (he == hipSuccess) he = hipMemsetAsync(d_top, 0, sizeof(uint32_t), c->stream);
            if (he != hipSuccess) { rc = fail_hip("mcx_profile_reduce", he); (void)ncclCommAbort(c->nccl); c->nccl = nullptr; return rc; }
            k_low_max<<<4096, 256, 0, c->stream>>>(plane(pF1), plane(pF2), n, d_top);
            ncclResult_t e = ncclAllReduce(d_top, d_top, 1, ncclUint32, ncclMax, c->nccl, c->stream);
            if (e != ncclSuccess) { (void)hipFree(d_top); return dead("ncclAllReduce", e); }
            he = hipMemcpyAsync(&top, d_top, sizeof top, hipMemcpyDeviceToHost, c->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
            (void)hipFree(d_top);
            if (he != hipSuccess) { rc = fail_hip("mcx_profile_reduce", he); (void)ncclCommAbort(c->nccl); c->nccl = nullptr; return rc; }
            strands_share = (uint64_t)top * (uint64_t)c->size <= 0xFFFFu;
        }
        const bool counters_share = c->size <= 16;
        struct Pair { int lo, hi, clamp; bool on; };
        const Pair pairs[4] = {{pA, pC, 1, counters_share}, {pG, pT, 1, counters_share}, {pF1, pR2, 0, strands_share}, {pF2, pR1, 0, strands_share}};
        for (const Pair &q : pairs) {
            if (q.on) {
                k_pack_planes<<<8192, 256, 0, c->stream>>>(plane(q.lo), plane(q.hi), n, q.clamp);
                if ((rc = reduce_plane(q.lo))) return rc;
                if (c->rank == root) k_unpack_planes<<<8192, 256, 0, c->stream>>>(plane(q.lo), plane(q.hi), n);
            } else {
                if ((rc = reduce_plane(q.lo)) || (rc = reduce_plane(q.hi))) return rc;
            }
        }
        if ((rc = reduce_plane(pMulti))) return rc;
        hipError_t he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) { rc = fail_hip("hipStreamSynchronize", he); (void)ncclCommAbort(c->nccl); c->nccl = nullptr; return rc; }
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
                for (int k = 0; k < kPlanes && rc == 0; k++) {
                    if (k == kReadCount) continue;
                    k_add_planes<<<4096, 256, 0, c->stream>>>(d_planes + (uint64_t)k * (uint64_t)G, (const uint32_t *)g.ptr[(size_t)r] + (uint64_t)k * (uint64_t)G, (uint64_t)G);
                    if ((he = hipGetLastError()) != hipSuccess) rc = fail_hip("k_add_planes", he);
                }
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
