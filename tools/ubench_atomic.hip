// tools/ubench_atomic.hip — what does MI355X sustain for scattered 4-byte atomic adds into a multi-GB array (the -vcf planes)?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_atomic.hip -o tools/ubench_atomic
//   tools/ubench_atomic [array MiB] [million updates]
// modes: 0 one update at a random position per lane          1 positions increasing with the lane number, one line apart or more
//        2 random position, two updates in the same 64-B line  3 random position, plain store instead of an atomic
//        4 random position, load only                           5 positions increasing, ~390 apart (reads in genome order)
//        6 random position, four updates in four arrays (planes) at the same position
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t s)
{
    s ^= s >> 33; s *= 0xff51afd7ed558ccdull; s ^= s >> 33; s *= 0xc4ceb9fe1a85ec53ull; s ^= s >> 33;
    return s;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_upd(uint32_t *a, uint64_t n, uint64_t updates, uint32_t *sink)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= updates) return;
    uint64_t p;
    if (MODE == 1) p = (t * 16) % n;
    else if (MODE == 5) p = (t * 390 + (mix(t) & 255)) % n;
    else p = mix(t + 12345) % n;
    if (MODE == 0 || MODE == 1 || MODE == 5) atomicAdd(a + p, 1u);
    else if (MODE == 2) { atomicAdd(a + p, 1u); atomicAdd(a + (p ^ 3), 0xFFFFFFFFu); }
    else if (MODE == 3) a[p] = (uint32_t)t;
    else if (MODE == 4) { if (a[p] == 0x12345u) sink[0] = 1; }
    else if (MODE == 6) { const uint64_t q = n / 4, b = p % q; atomicAdd(a + b, 1u); atomicAdd(a + q + b, 1u); atomicAdd(a + 2 * q + b, 1u); atomicAdd(a + 3 * q + b, 1u); }
}

template <int MODE>
static void run(uint32_t *a, uint64_t n, uint64_t updates, uint32_t *sink, const char *name, int per)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned blocks = (unsigned)((updates + 255) / 256);
    k_upd<MODE><<<blocks, 256>>>(a, n, updates / 16, sink);
    (void)hipEventRecord(e0);
    k_upd<MODE><<<blocks, 256>>>(a, n, updates, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %8.3f ms  %7.2f G lanes/s  %7.2f G updates/s\n", name, ms, updates / ms / 1e6, updates * per / ms / 1e6);
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? atol(argv[1]) : 12288;
    const uint64_t updates = (argc > 2 ? atol(argv[2]) : 40) * 1000000ull;
    const uint64_t n = mib * 1024 * 1024 / 4;
    uint32_t *a, *sink;
    (void)hipMalloc(&a, n * 4); (void)hipMalloc(&sink, 64);
    (void)hipMemset(a, 0, n * 4);
    printf("array %zu MiB, %llu M lanes\n", mib, (unsigned long long)(updates / 1000000));
    run<0>(a, n, updates, sink, "0 random position, one atomic add", 1);
    run<1>(a, n, updates, sink, "1 lane t at line t (a wave covers 64 consecutive lines)", 1);
    run<5>(a, n, updates, sink, "5 increasing positions ~390 apart", 1);
    run<2>(a, n, updates, sink, "2 random position, two atomics in one line", 2);
    run<6>(a, n, updates, sink, "6 random position, one atomic in each of four planes", 4);
    run<3>(a, n, updates, sink, "3 random position, plain 4-byte store", 1);
    run<4>(a, n, updates, sink, "4 random position, 4-byte load", 1);
    return 0;
}
