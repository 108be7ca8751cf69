// Probe: do CU-masked streams (hipExtStreamCreateWithCUMask) give real spatial partitioning on MI355X?
//   1. a compute-bound kernel on masks of 256 / 240 / 224 / 32 / 16 CUs: time should scale ~1/CUs;
//   2. the compute-bound kernel on 240 CUs and a second one on the other 16, launched together:
//      total time ~ max, not sum, if they really run side by side.
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_cumask tools/probe_cumask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void burn(double *out, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.000001;
    for (int i = 0; i < iters; ++i) { a = fma(a, b, 1e-9); b = fma(b, 0.999999, 1e-12); }
    if (a == 12345.678) out[0] = a + b;
}
__global__ void which_cu(unsigned *slots) {
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        atomicOr(&slots[(xcc & 0xf) * 8 + se], 1u << (sh * 16 + cu));
    }
}

static float run(hipStream_t s, int blocks, int iters, double *d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s);
    hipLaunchKernelGGL(burn, dim3(blocks), dim3(512), 0, s, d, iters);
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    double *d; CK(hipMalloc(&d, 64));
    unsigned *slots; CK(hipMalloc(&slots, 16 * 8 * 4));
    auto make = [&](int first, int count, hipStream_t *s) {
        std::vector<uint32_t> mask(8, 0);
        for (int c = first; c < first + count; ++c) mask[c >> 5] |= 1u << (c & 31);
        return hipExtStreamCreateWithCUMask(s, 8, mask.data());
    };
    const int iters = 200000;
    for (int n : {256, 240, 224, 128, 32, 16}) {
        hipStream_t s; CK(make(0, n, &s));
        run(s, 512, 1000, d);
        float ms = run(s, 512, iters, d);
        CK(hipMemset(slots, 0, 16 * 8 * 4));
        hipLaunchKernelGGL(which_cu, dim3(8192), dim3(64), 0, s, slots);
        CK(hipStreamSynchronize(s));
        unsigned h[128]; CK(hipMemcpy(h, slots, sizeof(h), hipMemcpyDeviceToHost));
        int cus = 0, xccs = 0;
        for (int x = 0; x < 16; ++x) { int c = 0; for (int se = 0; se < 8; ++se) c += __builtin_popcount(h[x * 8 + se]); cus += c; xccs += c > 0; }
        printf("mask first %3d CUs: burn %.3f ms; waves seen on %d CUs in %d XCCs\n", n, ms, cus, xccs);
        hipStreamDestroy(s);
    }
    hipStream_t big, small; CK(make(0, 240, &big)); CK(make(240, 16, &small));
    run(big, 512, 1000, d); run(small, 64, 1000, d);
    float t_big = run(big, 480, iters, d), t_small = run(small, 32, iters, d);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    CK(hipDeviceSynchronize());
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(burn, dim3(480), dim3(512), 0, big, d, iters);
    hipLaunchKernelGGL(burn, dim3(32), dim3(512), 0, small, d, iters);
    CK(hipDeviceSynchronize());
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float both; hipEventElapsedTime(&both, a, b);
    printf("240-CU stream alone %.3f ms, 16-CU stream alone %.3f ms, together %.3f ms\n", t_big, t_small, both);
    return 0;
}
