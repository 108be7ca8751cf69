// Diagnostic (not part of the product): accuracy of v_rsq_f64 and of refinement variants on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_math tools/probe_math.hip && /tmp/probe_math
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* y0, double* y1, double* y2, double* y3, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double y = __builtin_amdgcn_rsq(v);
    y0[i] = y;
    double t = v * y, e = fma(-t, y, 1.0);
    y1[i] = fma(y * e, 0.5, y);                       // plain Newton
    y2[i] = fma(y * e, fma(0.375, e, 0.5), y);        // + second-order term
    y3[i] = rsqrt(v);                                 // ocml
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n), c(n), d(n);
    for (int i = 0; i < n; ++i) x[i] = 0.01 * std::exp(12.0 * (i + 0.5) / n);   // 0.01 .. 1600
    double *dx, *d0, *d1, *d2, *d3;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, d3, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(d.data(), d3, n * 8, hipMemcpyDeviceToHost);
    long double m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    for (int i = 0; i < n; ++i) {
        long double r = 1.0L / sqrtl((long double)x[i]);
        m0 = fmaxl(m0, fabsl(a[i] - r) / r); m1 = fmaxl(m1, fabsl(b[i] - r) / r);
        m2 = fmaxl(m2, fabsl(c[i] - r) / r); m3 = fmaxl(m3, fabsl(d[i] - r) / r);
    }
    printf("max rel err: v_rsq_f64 %.3Le  newton %.3Le  newton2 %.3Le  ocml rsqrt %.3Le (eps=%.3e)\n", m0, m1, m2, m3, 2.22e-16);
    return 0;
}
