// Probe: sustained rate of v_mfma_f64_16x16x4_f64 (2048 flop per instruction) with 1..4 independent accumulators per wave
// and 1..3 waves per SIMD, against the fp64 vector peak (256 CUs x 4 SIMDs x 16 lanes x 2 flop x f).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void mfma_chains(double *out, int iters, double seed) {
    double4v acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = {seed, seed, seed, seed};
    double a = seed + threadIdx.x * 1e-9, b = 1.0 - 1e-9 * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 1234.5) out[0] = s;
}
template <int CH>
static void run(double *d, int blocks) {
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((mfma_chains<CH>), dim3(blocks), dim3(256), 0, 0, d, 100, 1e-30);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((mfma_chains<CH>), dim3(blocks), dim3(256), 0, 0, d, iters, 1e-30);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n_mfma = (double)CH * iters * blocks * 4;
    const double tf = n_mfma * 2048.0 / (ms * 1e-3) / 1e12;
    const double waves_per_simd = blocks * 4.0 / (256.0 * 4);
    printf("accumulators %d, %d workgroups of 4 waves (%.1f waves per SIMD): %.3f ms  %.1f TFLOP/s  -> %.1f cycles per instruction per SIMD at 2.4 GHz\n",
           CH, blocks, waves_per_simd, ms, tf, ms * 1e-3 * 2.4e9 / (n_mfma / 1024.0));
}
int main() {
    double *d; hipMalloc(&d, 64);
    run<1>(d, 256); run<2>(d, 256); run<4>(d, 256); run<4>(d, 512); run<4>(d, 768); run<1>(d, 768);
    return 0;
}
