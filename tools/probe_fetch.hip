// Diagnostic (not part of the product): calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the
// access widths the engine uses (8 B per lane coalesced reads; 16 B per lane for comparison).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_fetch tools/probe_fetch.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- /tmp/probe_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read8(const double* __restrict__ p, size_t n, double* out) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 1.2345e301) out[0] = acc;
}
__global__ void read16(const double2* __restrict__ p, size_t n, double* out) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 1.2345e301) out[0] = acc;
}
__global__ void write8(double* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0;
}
int main() {
    const size_t bytes = (size_t)2 << 30;   // 2 GiB: far beyond the 256 MiB Infinity Cache
    double *p, *out;
    hipMalloc(&p, bytes); hipMalloc(&out, 8);
    hipMemset(p, 0, bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(read8, dim3(4096), dim3(256), 0, 0, p, bytes / 8, out);
    hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const double2*)p, bytes / 16, out);
    hipLaunchKernelGGL(write8, dim3(4096), dim3(256), 0, 0, p, bytes / 8);
    hipDeviceSynchronize();
    printf("each kernel touches %zu bytes\n", bytes);
    return 0;
}
