// Microbenchmark: per-lane walks over pair-state-like records, record-major (AoS) against
// 64-pair interleaved (AoSoA, 16-byte granules).  hipcc --offload-arch=gfx950 -O3 layout_ubench.hip -o layout_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct U4 { uint32_t x, y, z, w; };
constexpr int kRec = 5120;             // bytes per pair record
constexpr int kG = kRec / 16;          // granules per record
template <bool SOA> __device__ __forceinline__ const U4 *at(const U4 *base, uint32_t pair, int g)
{
    if (SOA) return base + ((uint64_t)(pair >> 6) * kG + g) * 64 + (pair & 63);
    return base + (uint64_t)pair * kG + g;
}
// chain: header (4 granules) -> n cands (2 granules each, offset from header) -> 3 frags each (offset from cand) -> out
template <bool SOA, bool RND> __global__ void __launch_bounds__(256) k_walk(const U4 *st, uint32_t n, U4 *out)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    U4 h = *at<SOA>(st, p, 0);
    U4 h1 = *at<SOA>(st, p, 1);
    uint32_t acc = h1.x;
    const int nc = 1 + (h.x & 3);
    for (int c = 0; c < nc; c++) {
        const int cg = 4 + 2 * (RND ? ((h.y + c) & 15) : c);
        U4 a = *at<SOA>(st, p, cg), b = *at<SOA>(st, p, cg + 1);
        acc += b.x;
        const int nf = 2 + (a.x & 3);
        const int f0 = 64 + (RND ? (a.y & 63) : 5 * c);
        for (int f = 0; f < nf; f++) { U4 v = *at<SOA>(st, p, f0 + f); acc += v.x ^ v.w; }
    }
    U4 o; o.x = acc; o.y = h.z; o.z = h.w; o.w = 1;
    out[(uint64_t)p * 4] = o; out[(uint64_t)p * 4 + 1] = o; out[(uint64_t)p * 4 + 2] = o; out[(uint64_t)p * 4 + 3] = o;
}
__global__ void k_fill(U4 *p, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7);
        x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
        U4 v; v.x = x; v.y = x * 3u + 1u; v.z = x >> 3; v.w = x ^ 0x55u; p[i] = v;
    }
}
int main()
{
    const uint32_t n = 4u << 20;
    U4 *st, *out;
    hipMalloc(&st, (uint64_t)n * kRec); hipMalloc(&out, (uint64_t)n * 64);
    k_fill<<<65536, 256>>>(st, (uint64_t)n * kG);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 4; mode++) for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        if (mode == 0) k_walk<false, false><<<n / 256, 256>>>(st, n, out);
        if (mode == 1) k_walk<true, false><<<n / 256, 256>>>(st, n, out);
        if (mode == 2) k_walk<false, true><<<n / 256, 256>>>(st, n, out);
        if (mode == 3) k_walk<true, true><<<n / 256, 256>>>(st, n, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s %s rep %d: %.3f ms\n", (mode & 1) ? "interleaved" : "record-major", mode >= 2 ? "random-offsets" : "aligned-offsets", rep, ms);
    }
    return 0;
}
