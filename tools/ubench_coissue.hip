// ubench_coissue.hip -- does a 32-bit / transcendental VALU instruction cost issue time next to a stream of
// fp64 instructions on gfx950, or does it overlap with them?  Each kernel runs ITER x {8 independent
// v_fma_f64 + EXTRA} per wave at 4 waves per SIMD (16 per CU) on every CU; wall time per iteration from HIP
// events.  Measurement tool, not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 20000;

#define FMA8 "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n" \
             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"

#define KERNEL(NAME, BODY)                                                                        \
__global__ __launch_bounds__(1024) void NAME(double *out) {                                      \
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + .1, a2 = a0 + .2, a3 = a0 + .3,               \
           a4 = a0 + .4, a5 = a0 + .5, a6 = a0 + .6, a7 = a0 + .7;                               \
    double b = 1.0000000001, c = 1e-9, d0 = a0 + 3.0, d1 = a0 + 4.0;                              \
    float f0 = threadIdx.x * 1e-3f + 1.0f, f1 = f0 + .1f, f2 = f0 + .2f, f3 = f0 + .3f;           \
    float g = 1.0000001f, h = 1e-6f;                                                              \
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;                                  \
    for (int i = 0; i < ITER; ++i) {                                                              \
        asm volatile(BODY                                                                         \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(b), "v"(c));                                                           \
        asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(d0), "+v"(d1), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(g), "v"(h)); \
    }                                                                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + d0 + d1 + i0 + i1 + i2 + i3; \
}

// the EXTRA instructions work on their own registers through a second asm block with explicit operands
#define KERNEL2(NAME, BODY8, EXTRA)                                                               \
__global__ __launch_bounds__(1024) void NAME(double *out) {                                      \
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + .1, a2 = a0 + .2, a3 = a0 + .3,               \
           a4 = a0 + .4, a5 = a0 + .5, a6 = a0 + .6, a7 = a0 + .7;                               \
    double b = 1.0000000001, c = 1e-9, d0 = a0 + 3.0, d1 = a0 + 4.0;                              \
    float f0 = threadIdx.x * 1e-3f + 1.0f, f1 = f0 + .1f, f2 = f0 + .2f, f3 = f0 + .3f;           \
    float g = 1.0000001f, h = 1e-6f;                                                              \
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;                                  \
    for (int i = 0; i < ITER; ++i) {                                                              \
        asm volatile(BODY8 EXTRA                                                                  \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                       "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(d0), "+v"(d1), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) \
                     : "v"(b), "v"(c), "v"(g), "v"(h));                                           \
    }                                                                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + d0 + d1 + i0 + i1 + i2 + i3; \
}
// operand map for KERNEL2: %0-%7 a (f64), %8-%11 f (f32), %12 %13 d (f64), %14-%17 i (int), %18 b, %19 c, %20 g, %21 h
#define F8 "v_fma_f64 %0, %0, %18, %19\n v_fma_f64 %1, %1, %18, %19\n v_fma_f64 %2, %2, %18, %19\n v_fma_f64 %3, %3, %18, %19\n" \
           "v_fma_f64 %4, %4, %18, %19\n v_fma_f64 %5, %5, %18, %19\n v_fma_f64 %6, %6, %18, %19\n v_fma_f64 %7, %7, %18, %19\n"
#define DPP4 "v_mov_b32_dpp %14, %15 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %15, %16 wave_ror:1 row_mask:0xf bank_mask:0xf\n" \
             "v_mov_b32_dpp %16, %17 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %17, %14 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
#define FMA32_4 "v_fma_f32 %8, %8, %20, %21\n v_fma_f32 %9, %9, %20, %21\n v_fma_f32 %10, %10, %20, %21\n v_fma_f32 %11, %11, %20, %21\n"
#define ADDU4 "v_add_u32 %14, %14, %15\n v_add_u32 %15, %15, %16\n v_add_u32 %16, %16, %17\n v_add_u32 %17, %17, %14\n"
#define RCP64 "v_rcp_f64 %12, %12\n"
#define RCP32 "v_rcp_f32 %8, %8\n"
#define CVT2 "v_cvt_f32_f64 %9, %12\n v_cvt_f64_f32 %12, %10\n"
#define RCPVIA32 "v_cvt_f32_f64 %9, %12\n v_rcp_f32 %9, %9\n v_cvt_f64_f32 %12, %9\n"

KERNEL2(k_fma8, F8, "")
KERNEL2(k_fma8_dpp4, F8, DPP4)
KERNEL2(k_fma8_dpp8, F8, DPP4 DPP4)
KERNEL2(k_fma8_fma32x4, F8, FMA32_4)
KERNEL2(k_fma8_addu4, F8, ADDU4)
KERNEL2(k_fma8_rcp64, F8, RCP64)
KERNEL2(k_fma8_rcp32, F8, RCP32)
KERNEL2(k_fma8_cvt2, F8, CVT2)
KERNEL2(k_fma8_rcpvia32, F8, RCPVIA32)
KERNEL2(k_dpp4, "", DPP4)
KERNEL2(k_dpp8, "", DPP4 DPP4)
KERNEL2(k_fma32x4, "", FMA32_4)
KERNEL2(k_rcp64, "", RCP64)
KERNEL2(k_rcp32, "", RCP32)
KERNEL2(k_rcpvia32, "", RCPVIA32)
KERNEL2(k_fma16, F8 F8, "")

typedef void (*kern_t)(double *);
struct Entry { const char *name; kern_t k; };
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    Entry tab[] = {{"8 fma_f64", k_fma8}, {"16 fma_f64", k_fma16}, {"8 fma_f64 + 4 dpp mov", k_fma8_dpp4}, {"8 fma_f64 + 8 dpp mov", k_fma8_dpp8},
                   {"8 fma_f64 + 4 fma_f32", k_fma8_fma32x4}, {"8 fma_f64 + 4 add_u32", k_fma8_addu4},
                   {"8 fma_f64 + 1 rcp_f64", k_fma8_rcp64}, {"8 fma_f64 + 1 rcp_f32", k_fma8_rcp32},
                   {"8 fma_f64 + cvt,cvt", k_fma8_cvt2}, {"8 fma_f64 + cvt,rcp_f32,cvt", k_fma8_rcpvia32},
                   {"4 dpp mov", k_dpp4}, {"8 dpp mov", k_dpp8}, {"4 fma_f32", k_fma32x4}, {"1 rcp_f64", k_rcp64}, {"1 rcp_f32", k_rcp32},
                   {"cvt,rcp_f32,cvt", k_rcpvia32}};
    const int blocks = 256, threads = 1024;
    double *out;
    CK(hipMalloc(&out, sizeof(double) * blocks * threads));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-32s %12s   (ns per loop iteration; 4 waves/SIMD, all CUs)\n", "per iteration and wave", "ns/iter");
    for (auto &t : tab) {
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(threads), 0, 0, out);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(threads), 0, 0, out);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%-32s %12.3f\n", t.name, best * 1e6 / ITER);
    }
    return 0;
}
