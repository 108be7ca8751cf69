// Probe: sustained fp64 FMA rate of the whole chip with the pair sweep's launch shape (2 workgroups of 8 waves
// per CU = 4 waves per SIMD), 1..8 independent dependency chains per lane.  Peak = 256 CUs x 4 SIMDs x 16
// lanes x 2 flop x f; the measured rate gives the clock the chip actually sustains under fp64 load.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int BLOCK = 512>
__global__ __launch_bounds__(BLOCK) void fma_chains(double *out, int iters, double seed) {
    double a[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = seed + threadIdx.x * 1e-6 + c;
    const double m = 1.0 - 1e-9, k = 1e-7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) a[c] = fma(a[c], m, k);
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += a[c];
    if (s == 1234.5) out[0] = s;
}
template <int CH, int BLOCK = 512>
static void run(double *d, int blocks) {
    const int iters = 100000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((fma_chains<CH, BLOCK>), dim3(blocks), dim3(BLOCK), 0, 0, d, 1000, 1.0);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((fma_chains<CH, BLOCK>), dim3(blocks), dim3(BLOCK), 0, 0, d, iters, 1.0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flops = 2.0 * CH * (double)iters * blocks * BLOCK;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("chains %d, %d x %d threads: %.3f ms  %.1f TFLOP/s  -> implied clock %.2f GHz (of 256x4x16 lanes)\n", CH, blocks, BLOCK, ms, tf,
           tf * 1e12 / (256.0 * 4 * 16 * 2) / 1e9);
}
int main() {
    double *d; hipMalloc(&d, 64);
    run<1>(d, 512); run<2>(d, 512); run<3>(d, 512); run<4>(d, 512); run<8>(d, 512);
    run<3>(d, 256); run<8>(d, 256);
    // 3 waves per SIMD (2 workgroups of 6 waves per CU) with 6 chains; 2 waves per SIMD with 6
    run<6, 384>(d, 512); run<6, 512>(d, 256); run<6, 512>(d, 512); run<5, 512>(d, 512);
    run<6, 256>(d, 768); run<6, 256>(d, 1024); run<3, 256>(d, 1024); run<6, 256>(d, 512);
    return 0;
}
