// Probe: dynamic LDS per workgroup beyond the default 64 KB (hipFuncAttributeMaxDynamicSharedMemorySize) on gfx950: 48-160 KB allocated,
// written and read back.  profiles/r05/probe_lds_big.txt: all sizes up to 160 KB launch and run.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, int n) {
    extern __shared__ double s[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = i * 0.5;
    __syncthreads();
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += s[n - 1 - i];
    if (threadIdx.x == 0) out[blockIdx.x] = a + s[n - 1];
}
int main() {
    int v = 0, dev = 0;
    hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev); printf("MaxSharedMemoryPerBlock %d\n", v);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, dev); printf("sharedMemPerBlock %zu optin %zu perMultiprocessor %zu\n", p.sharedMemPerBlock, p.sharedMemPerBlockOptin, p.sharedMemPerMultiprocessor);
    double *d; hipMalloc(&d, 1024 * 8);
    for (size_t kb : {48, 64, 96, 128, 144, 160}) {
        size_t bytes = kb * 1024;
        hipError_t e1 = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        hipLaunchKernelGGL(k, dim3(512), dim3(256), bytes, 0, d, (int)(bytes / 8));
        hipError_t e2 = hipGetLastError(); hipError_t e3 = hipDeviceSynchronize();
        double h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("%zu KB: setattr %s launch %s sync %s out %.1f\n", kb, hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e3), h);
    }
    return 0;
}
