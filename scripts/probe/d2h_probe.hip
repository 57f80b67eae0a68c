// Which engine carries a device-to-host copy, and what it costs a streaming kernel that runs beside it.
//   hipcc --offload-arch=gfx950 -O2 -o d2h_probe d2h_probe.hip ; rocprofv3 --kernel-trace --memory-copy-trace ... -- ./d2h_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int CASE> __global__ void k_case(int *p) { if (p && threadIdx.x == 999) p[0] = CASE; }
__global__ void k_stream(const uint4 *a, uint4 *b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = a[i]; if ((i & 3) == 0) b[i >> 2] = v;
    }
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    size_t out = 256u << 20, in = 320u << 20, big = 1200u << 20;
    uint8_t *d_out, *d_in, *d_a, *d_b, *h_out, *h_in, *h_reg;
    CK(hipMalloc(&d_out, out)); CK(hipMalloc(&d_in, in)); CK(hipMalloc(&d_a, big)); CK(hipMalloc(&d_b, big / 4));
    CK(hipHostMalloc(&h_out, out)); CK(hipHostMalloc(&h_in, in));
    h_reg = (uint8_t *)aligned_alloc(4096, out); for (size_t i = 0; i < out; i += 4096) h_reg[i] = 1;
    CK(hipHostRegister(h_reg, out, hipHostRegisterDefault));
    CK(hipMemset(d_a, 1, big)); CK(hipMemset(d_out, 2, out));
    hipStream_t s1, s2, s3; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto stream_kernel = [&](hipStream_t s) { k_stream<<<4096, 256, 0, s>>>((const uint4 *)d_a, (uint4 *)d_b, big / 16); };
    auto timed = [&](const char *what, auto &&side) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipDeviceSynchronize());
            double t0 = now();
            side();
            CK(hipEventRecord(e0, s3)); stream_kernel(s3); CK(hipEventRecord(e1, s3));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("%-58s streaming kernel %.2f ms, all %.2f ms\n", what, ms, now() - t0);
        }
    };
    k_case<0><<<1, 64, 0, s1>>>(nullptr);
    timed("0 nothing beside it", [&] {});
    k_case<1><<<1, 64, 0, s1>>>(nullptr);
    timed("1 D2H 256 MB to hipHostMalloc, alone on its stream", [&] { CK(hipMemcpyAsync(h_out, d_out, out, hipMemcpyDeviceToHost, s1)); });
    k_case<2><<<1, 64, 0, s1>>>(nullptr);
    timed("2 a kernel, then the D2H on the same stream", [&] { k_case<20><<<1, 64, 0, s1>>>(nullptr); CK(hipMemcpyAsync(h_out, d_out, out, hipMemcpyDeviceToHost, s1)); });
    k_case<3><<<1, 64, 0, s1>>>(nullptr);
    timed("3 H2D 320 MB alone", [&] { CK(hipMemcpyAsync(d_in, h_in, in, hipMemcpyHostToDevice, s2)); });
    k_case<4><<<1, 64, 0, s1>>>(nullptr);
    timed("4 H2D on one stream, kernel + D2H on another", [&] { CK(hipMemcpyAsync(d_in, h_in, in, hipMemcpyHostToDevice, s2)); k_case<20><<<1, 64, 0, s1>>>(nullptr); CK(hipMemcpyAsync(h_out, d_out, out, hipMemcpyDeviceToHost, s1)); });
    k_case<5><<<1, 64, 0, s1>>>(nullptr);
    timed("5 D2H to hipHostRegister memory", [&] { CK(hipMemcpyAsync(h_reg, d_out, out, hipMemcpyDeviceToHost, s1)); });
    k_case<6><<<1, 64, 0, s1>>>(nullptr);
    timed("6 D2H in 8 pieces of 32 MB", [&] { for (int i = 0; i < 8; i++) CK(hipMemcpyAsync(h_out + (size_t)i * (32u << 20), d_out + (size_t)i * (32u << 20), 32u << 20, hipMemcpyDeviceToHost, s1)); });
    k_case<7><<<1, 64, 0, s1>>>(nullptr);
    timed("7 D2H by hipMemcpyDtoHAsync", [&] { CK(hipMemcpyDtoHAsync(h_out, (hipDeviceptr_t)d_out, out, s1)); });

    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    auto busy = [&](hipStream_t s, int n) { for (int i = 0; i < n; i++) stream_kernel(s); };
    k_case<8><<<1, 64, 0, s1>>>(nullptr);
    timed("8 6 ms of kernels, then the D2H, same stream", [&] { busy(s1, 20); CK(hipMemcpyAsync(h_out, d_out, out, hipMemcpyDeviceToHost, s1)); });
    k_case<9><<<1, 64, 0, s1>>>(nullptr);
    timed("9 6 ms of kernels on another stream, event, wait, D2H", [&] { busy(s2, 20); CK(hipEventRecord(ev, s2)); CK(hipStreamWaitEvent(s1, ev, 0)); CK(hipMemcpyAsync(h_out, d_out, out, hipMemcpyDeviceToHost, s1)); });
    hipStream_t many[12]; for (auto &m : many) CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking));
    k_case<10><<<1, 64, 0, s1>>>(nullptr);
    timed("10 twelve more streams with kernels, then as 9", [&] { for (auto &m : many) busy(m, 4); busy(s2, 20); CK(hipEventRecord(ev, s2)); CK(hipStreamWaitEvent(s1, ev, 0)); CK(hipMemcpyAsync(h_out, d_out, out, hipMemcpyDeviceToHost, s1)); });
    k_case<11><<<1, 64, 0, s1>>>(nullptr);
    timed("11 as 10, D2H on the last of the twelve", [&] { for (auto &m : many) busy(m, 4); busy(s2, 20); CK(hipEventRecord(ev, s2)); CK(hipStreamWaitEvent(many[11], ev, 0)); CK(hipMemcpyAsync(h_out, d_out, out, hipMemcpyDeviceToHost, many[11])); });
    k_case<12><<<1, 64, 0, s1>>>(nullptr);
    timed("12 as 11 with the H2D in flight and a second D2H behind", [&] { CK(hipMemcpyAsync(d_in, h_in, in, hipMemcpyHostToDevice, many[3])); for (auto &m : many) busy(m, 4); busy(s2, 20); CK(hipEventRecord(ev, s2)); CK(hipStreamWaitEvent(many[11], ev, 0)); CK(hipMemcpyAsync(h_out, d_out, out - (64u << 20), hipMemcpyDeviceToHost, many[11])); CK(hipMemcpyAsync(h_out + out - (64u << 20), d_out + out - (64u << 20), 64u << 20, hipMemcpyDeviceToHost, many[11])); });
    CK(hipDeviceSynchronize());
    return 0;
}
