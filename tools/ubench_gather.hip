// tools/ubench_gather.hip — what does MI355X sustain for DEPENDENT random 64-byte block reads?
// (the access pattern of the FM-index walk: every lane chases its own chain of blocks)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_gather.hip -o tools/ubench_gather
//   tools/ubench_gather [buffer MiB] [steps per lane] [quick: modes 2, 4, 7 at 4096 blocks — for sweeps over the buffer size]
// modes: 0 lane-per-block 4 x 16 B   1 same, non-temporal   2 four lanes per block (one 16 B load each)
//        3 lane-per-block, 128-B blocks (8 x 16 B)            4 lane-per-block, first 32 B only
//        5 two independent chains per lane (ILP)
//        7 lane-per-block, first 16 B only (one 16-byte load per line and lane: the seeding walk's record fetch)
//        6 one chain per lane, served by the wave four lanes per block: four rounds of 16 requests, the address pulled from
//          the requesting lane and the result pushed back with ds_bpermute (what k_seed's FM step would do)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct alignas(16) U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ uint64_t mix(uint64_t s, uint32_t v)
{
    s ^= v; s *= 0x9E3779B97F4A7C15ull; s ^= s >> 29;
    return s;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_chase(const U4 *buf, uint64_t n_blocks, int steps, uint64_t *out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = 0x1234567ull * (tid + 1), s2 = 0x7654321ull * (tid + 7);
    uint32_t acc = 0;
    if (MODE == 2) s = 0x1234567ull * ((tid >> 2) + 1); // the four lanes of a quad share a chain
    for (int i = 0; i < steps; i++) {
        if (MODE == 0 || MODE == 4 || MODE == 5) {
            const U4 *p = buf + (s % n_blocks) * 4;
            U4 a = p[0], b = p[1];
            uint32_t v = a.x ^ a.w ^ b.y;
            if (MODE != 4) { U4 c = p[2], d = p[3]; v ^= c.z ^ d.x; }
            acc += v; s = mix(s, v);
            if (MODE == 5) {
                const U4 *q = buf + (s2 % n_blocks) * 4;
                U4 a2 = q[0], b2 = q[1], c2 = q[2], d2 = q[3];
                uint32_t v2 = a2.x ^ a2.w ^ b2.y ^ c2.z ^ d2.x;
                acc += v2; s2 = mix(s2, v2);
            }
        } else if (MODE == 1) {
            const uint32_t *p = (const uint32_t *)(buf + (s % n_blocks) * 4);
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 16; k += 4) {
                v ^= __builtin_nontemporal_load(p + k) ^ __builtin_nontemporal_load(p + k + 3);
            }
            acc += v; s = mix(s, v);
        } else if (MODE == 2) {
            const U4 *p = buf + (s % n_blocks) * 4 + (threadIdx.x & 3);
            U4 a = *p;
            uint32_t v = a.x ^ a.w;
            v ^= __shfl_xor(v, 1, 64); v ^= __shfl_xor(v, 2, 64);
            acc += v; s = mix(s, v);
        } else if (MODE == 6) {
            const int lane = threadIdx.x & 63;
            const uint64_t blk = s % n_blocks;
            const uint32_t blo = (uint32_t)blk, bhi = (uint32_t)(blk >> 32);
            U4 d[4];
#pragma unroll
            for (int r = 0; r < 4; r++) { // all four rounds' fetches are issued before any is waited for
                const int src = 16 * r + (lane >> 2);
                const uint64_t b = (uint64_t)(uint32_t)__shfl((int)blo, src, 64) | ((uint64_t)(uint32_t)__shfl((int)bhi, src, 64) << 32);
                d[r] = buf[b * 4 + (lane & 3)];
            }
            uint32_t v = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uint32_t t = d[r].x ^ d[r].w;
                t ^= __shfl_xor((int)t, 1, 64); t ^= __shfl_xor((int)t, 2, 64);
                const uint32_t got = (uint32_t)__shfl((int)t, 4 * (lane & 15), 64); // the requester takes its quad's result
                if ((lane >> 4) == r) v = got;
            }
            acc += v; s = mix(s, v);
        } else if (MODE == 7) {
            const U4 a = buf[(s % n_blocks) * 4];
            const uint32_t v = a.x ^ a.w;
            acc += v; s = mix(s, v);
        } else if (MODE == 3) {
            const U4 *p = buf + (s % (n_blocks / 2)) * 8;
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) { U4 a = p[k]; v ^= a.x ^ a.w; }
            acc += v; s = mix(s, v);
        }
    }
    out[tid] = s + acc + s2;
}

__global__ void k_fill(uint32_t *p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7);
}

template <int MODE>
static void run(const U4 *buf, uint64_t n_blocks, int steps, uint64_t *out, int blocks, const char *name, double bytes_per_step)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_chase<MODE><<<blocks, 256>>>(buf, n_blocks, 8, out);
    hipEventRecord(e0);
    k_chase<MODE><<<blocks, 256>>>(buf, n_blocks, steps, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double lanes = (double)blocks * 256;
    double chains = MODE == 2 ? lanes / 4 : (MODE == 5 ? lanes * 2 : lanes);
    double blk = chains * steps;
    printf("%-44s blocks=%5d  %8.3f ms  %7.2f Gblock/s  %7.1f GB/s useful\n", name, blocks, ms, blk / ms / 1e6, blk * bytes_per_step / ms / 1e6);
}

int main(int argc, char **argv)
{
    size_t mib = argc > 1 ? atol(argv[1]) : 4096;
    int steps = argc > 2 ? atoi(argv[2]) : 512;
    uint64_t n_blocks = mib * 1024 * 1024 / 64;
    U4 *buf; uint64_t *out;
    hipMalloc(&buf, n_blocks * 64);
    hipMalloc(&out, 64ull << 20);
    if (!buf || !out) { printf("allocation of %zu MiB failed\n", mib); return 1; }
    k_fill<<<8192, 256>>>((uint32_t *)buf, n_blocks * 16);
    hipDeviceSynchronize();
    printf("buffer %zu MiB, %d dependent steps per chain\n", mib, steps);
    if (argc > 3) {
        run<2>(buf, n_blocks, steps, out, 4096, "2 quad-per-block 1x16B per lane", 64);
        run<4>(buf, n_blocks, steps, out, 4096, "4 lane-per-block first 32B only", 32);
        run<7>(buf, n_blocks, steps, out, 4096, "7 lane-per-block first 16B only", 16);
        run<7>(buf, n_blocks, steps, out, 16384, "7 lane-per-block first 16B only", 16);
        return 0;
    }
    for (int blocks : {1024, 2048, 4096, 8192}) {
        run<0>(buf, n_blocks, steps, out, blocks, "0 lane-per-block 4x16B", 64);
        run<1>(buf, n_blocks, steps, out, blocks, "1 lane-per-block non-temporal dwords", 64);
        run<2>(buf, n_blocks, steps, out, blocks, "2 quad-per-block 1x16B per lane", 64);
        run<3>(buf, n_blocks, steps, out, blocks, "3 lane-per-128B-block 8x16B", 128);
        run<4>(buf, n_blocks, steps, out, blocks, "4 lane-per-block first 32B only", 32);
        run<5>(buf, n_blocks, steps, out, blocks, "5 two chains per lane 4x16B", 64);
        run<6>(buf, n_blocks, steps, out, blocks, "6 lane-per-chain, quad-served in 4 rounds", 64);
        run<7>(buf, n_blocks, steps, out, blocks, "7 lane-per-block first 16B only", 16);
    }
    return 0;
}
