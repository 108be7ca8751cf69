// Probe: issue cost of the instruction kinds in the pair sweep's inner loop, at the sweep's launch shape
// (2 workgroups of 8 waves per CU = 4 waves per SIMD), 8 independent chains per lane, relative to v_fma_f64.
// Tells which instructions are worth removing from the loop (quarter-rate fp64 ops, LDS reads, conversions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define CH 8
#define BODY8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)

#define KERNEL(NAME, STMT)                                                                          \
    __global__ __launch_bounds__(512) void NAME(double *out, int iters, double seed, const char *tab) { \
        extern __shared__ __attribute__((aligned(16))) char lds[];                                  \
        for (int i = threadIdx.x; i < 2048; i += 512) reinterpret_cast<double2 *>(lds)[i] = make_double2(1.0 + i * 1e-9, 0.5); \
        __syncthreads();                                                                            \
        double a[CH]; float f[CH]; int u[CH]; double2 d2[CH];                                       \
        _Pragma("unroll") for (int c = 0; c < CH; ++c) { a[c] = seed + threadIdx.x * 1e-6 + c; f[c] = (float)a[c]; u[c] = (threadIdx.x * 37 + c * 101) & 1023; d2[c] = make_double2(0, 0); } \
        const double m = 1.0 - 1e-9, k = 1e-7;                                                      \
        const double magic = 6755399441055744.0;                                                    \
        (void)m; (void)k; (void)magic; (void)tab;                                                   \
        for (int i = 0; i < iters; ++i) { BODY8(STMT) }                                             \
        double s = 0.0;                                                                             \
        _Pragma("unroll") for (int c = 0; c < CH; ++c) s += a[c] + f[c] + u[c] + d2[c].x + d2[c].y; \
        if (s == 1234.5) out[0] = s;                                                                \
    }

#define S_FMA(c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(k));
#define S_ADD(c) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(k));
#define S_MUL(c) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[c]) : "v"(m));
#define S_RNDNE(c) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[c]));
#define S_MAGIC(c) asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, -%1" : "+v"(a[c]) : "v"(magic));
#define S_CVT(c) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[c]) : "v"(f[c]));
#define S_AND(c) asm volatile("v_and_b32 %0, 0x3fff, %0" : "+v"(u[c]));
#define S_MAD24(c) asm volatile("v_mad_u32_u24 %0, %0, 48, %1" : "+v"(u[c]) : "v"(u[(c + 1) & 7]));
#define S_MOV64(c) asm volatile("v_mov_b64 %0, %1" : "=v"(a[c]) : "v"(a[(c + 1) & 7]));
#define S_CMPSEL(c) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(u[c]) : "v"(a[c]), "v"(k), "v"(u[(c + 1) & 7]) : "vcc");
#define S_RCP(c) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[c]));
#define S_RSQ(c) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[c]));
#define S_LDS128_RAND(c) asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(d2[c]) : "v"(((u[c] * 48) & 0x7ff0)));
#define S_LDS128_SAME(c) asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(d2[c]) : "v"((c * 48) & 0x7ff0));
#define S_LDS128_RAND_NW(c) asm volatile("ds_read_b128 %0, %1" : "=v"(d2[c]) : "v"(((u[c] * 48) & 0x7ff0)));
#define S_LDS64_RAND_NW(c) asm volatile("ds_read_b64 %0, %1" : "=v"(a[c]) : "v"(((u[c] * 40) & 0x7ff8)));
#define S_FRACT(c) asm volatile("v_fract_f64 %0, %0" : "+v"(a[c]));
#define S_FMA32(c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(f[(c + 1) & 7]), "v"(f[(c + 2) & 7]));
#define S_PKFMA32(c) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(a[(c + 1) & 7]));

KERNEL(k_fma, S_FMA)
KERNEL(k_add, S_ADD)
KERNEL(k_mul, S_MUL)
KERNEL(k_rndne, S_RNDNE)
KERNEL(k_magic, S_MAGIC)
KERNEL(k_cvt, S_CVT)
KERNEL(k_and, S_AND)
KERNEL(k_mad24, S_MAD24)
KERNEL(k_mov64, S_MOV64)
KERNEL(k_cmpsel, S_CMPSEL)
KERNEL(k_rcp, S_RCP)
KERNEL(k_rsq, S_RSQ)
KERNEL(k_lds128_rand, S_LDS128_RAND)
KERNEL(k_lds128_same, S_LDS128_SAME)
KERNEL(k_lds128_rand_nw, S_LDS128_RAND_NW)
KERNEL(k_lds64_rand_nw, S_LDS64_RAND_NW)
KERNEL(k_fract, S_FRACT)
KERNEL(k_fma32, S_FMA32)
KERNEL(k_pkfma32, S_PKFMA32)

typedef void (*kern_t)(double *, int, double, const char *);
static double run(const char *name, kern_t kfn, double *d, int iters, double ref_ms, int per_stmt) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kfn, dim3(512), dim3(512), 32768, 0, d, 100, 1.0, (const char *)nullptr);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(kfn, dim3(512), dim3(512), 32768, 0, d, iters, 1.0, (const char *)nullptr);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD: 512 blocks x 8 waves / 1024 SIMDs = 4 waves per SIMD
    const double winstr = (double)iters * CH * per_stmt * 4.0;
    printf("%-18s %8.3f ms  %6.2f ns per wave-instr per SIMD  %5.2f x v_fma_f64\n", name, ms, ms * 1e6 / winstr,
           ref_ms > 0 ? (ms / per_stmt) / ref_ms : 1.0);
    return ms;
}
int main() {
    double *d; hipMalloc(&d, 64);
    const int iters = 20000;
    run("v_fma_f64(warm)", k_fma, d, iters, 0, 1);
    const double ref = run("v_fma_f64", k_fma, d, iters, 0, 1);
    run("v_add_f64", k_add, d, iters, ref, 1);
    run("v_mul_f64", k_mul, d, iters, ref, 1);
    run("v_rndne_f64", k_rndne, d, iters, ref, 1);
    run("magic rint(2 add)", k_magic, d, iters, ref, 2);
    run("v_fract_f64", k_fract, d, iters, ref, 1);
    run("v_cvt_f64_f32", k_cvt, d, iters, ref, 1);
    run("v_and_b32", k_and, d, iters, ref, 1);
    run("v_mad_u32_u24", k_mad24, d, iters, ref, 1);
    run("v_mov_b64", k_mov64, d, iters, ref, 1);
    run("cmp_f64+cndmask", k_cmpsel, d, iters, ref, 2);
    run("v_rcp_f64", k_rcp, d, iters, ref, 1);
    run("v_rsq_f64", k_rsq, d, iters, ref, 1);
    run("v_fma_f32", k_fma32, d, iters, ref, 1);
    run("v_pk_fma_f32", k_pkfma32, d, iters, ref, 1);
    run("ds_read_b128 rand+w", k_lds128_rand, d, iters, ref, 1);
    run("ds_read_b128 same+w", k_lds128_same, d, iters, ref, 1);
    run("ds_read_b128 rand", k_lds128_rand_nw, d, iters, ref, 1);
    run("ds_read_b64 rand", k_lds64_rand_nw, d, iters, ref, 1);
    return 0;
}
