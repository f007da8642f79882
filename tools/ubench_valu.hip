// ubench_valu.hip -- instruction-throughput microbenchmark for the gfx950 VALU ops the LJ pair
// kernel is built from (fp64 add/mul/fma/rndne/rcp/cmp, DPP wave rotate, ...).  Measurement tool,
// not product code.  Each kernel issues ITER x 8 independent instructions per wave and reports
// shader cycles (s_memtime) per wave-instruction for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITER = 4096;

#define BENCH_F64_3(NAME, INSTR)                                                              \
__global__ void NAME(double *out, long long *cyc) {                                          \
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + .1, a2 = a0 + .2, a3 = a0 + .3,           \
           a4 = a0 + .4, a5 = a0 + .5, a6 = a0 + .6, a7 = a0 + .7;                           \
    double b = 1.0000000001, c = 1e-9;                                                        \
    long long t0 = __builtin_amdgcn_s_memtime();                                              \
    for (int i = 0; i < ITER; ++i) {                                                          \
        asm volatile(INSTR " %0, %0, %8, %9\n" INSTR " %1, %1, %8, %9\n"                      \
                     INSTR " %2, %2, %8, %9\n" INSTR " %3, %3, %8, %9\n"                      \
                     INSTR " %4, %4, %8, %9\n" INSTR " %5, %5, %8, %9\n"                      \
                     INSTR " %6, %6, %8, %9\n" INSTR " %7, %7, %8, %9\n"                      \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(b), "v"(c));                                                       \
    }                                                                                         \
    long long t1 = __builtin_amdgcn_s_memtime();                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;       \
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
}

#define BENCH_F64_2(NAME, INSTR)                                                              \
__global__ void NAME(double *out, long long *cyc) {                                          \
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + .1, a2 = a0 + .2, a3 = a0 + .3,           \
           a4 = a0 + .4, a5 = a0 + .5, a6 = a0 + .6, a7 = a0 + .7;                           \
    double b = 1.0000000001;                                                                  \
    long long t0 = __builtin_amdgcn_s_memtime();                                              \
    for (int i = 0; i < ITER; ++i) {                                                          \
        asm volatile(INSTR " %0, %0, %8\n" INSTR " %1, %1, %8\n"                              \
                     INSTR " %2, %2, %8\n" INSTR " %3, %3, %8\n"                              \
                     INSTR " %4, %4, %8\n" INSTR " %5, %5, %8\n"                              \
                     INSTR " %6, %6, %8\n" INSTR " %7, %7, %8\n"                              \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(b));                                                               \
    }                                                                                         \
    long long t1 = __builtin_amdgcn_s_memtime();                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;       \
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
}

#define BENCH_F64_1(NAME, INSTR)                                                              \
__global__ void NAME(double *out, long long *cyc) {                                          \
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + .1, a2 = a0 + .2, a3 = a0 + .3,           \
           a4 = a0 + .4, a5 = a0 + .5, a6 = a0 + .6, a7 = a0 + .7;                           \
    long long t0 = __builtin_amdgcn_s_memtime();                                              \
    for (int i = 0; i < ITER; ++i) {                                                          \
        asm volatile(INSTR " %0, %0\n" INSTR " %1, %1\n" INSTR " %2, %2\n" INSTR " %3, %3\n"  \
                     INSTR " %4, %4\n" INSTR " %5, %5\n" INSTR " %6, %6\n" INSTR " %7, %7\n"  \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); \
    }                                                                                         \
    long long t1 = __builtin_amdgcn_s_memtime();                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;       \
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
}

#define BENCH_B32_RAW(NAME, BODY)                                                             \
__global__ void NAME(double *out, long long *cyc) {                                          \
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + .1f, a2 = a0 + .2f, a3 = a0 + .3f,       \
          a4 = a0 + .4f, a5 = a0 + .5f, a6 = a0 + .6f, a7 = a0 + .7f;                        \
    float b = 1.0000001f, c = 1e-6f;                                                          \
    long long t0 = __builtin_amdgcn_s_memtime();                                              \
    for (int i = 0; i < ITER; ++i) {                                                          \
        asm volatile(BODY                                                                     \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(b), "v"(c) : "vcc", "scc");                                               \
    }                                                                                         \
    long long t1 = __builtin_amdgcn_s_memtime();                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;       \
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
}

#define REP8_3(I) I " %0, %0, %8, %9\n" I " %1, %1, %8, %9\n" I " %2, %2, %8, %9\n" I " %3, %3, %8, %9\n" \
                  I " %4, %4, %8, %9\n" I " %5, %5, %8, %9\n" I " %6, %6, %8, %9\n" I " %7, %7, %8, %9\n"
#define REP8_2(I) I " %0, %0, %8\n" I " %1, %1, %8\n" I " %2, %2, %8\n" I " %3, %3, %8\n" \
                  I " %4, %4, %8\n" I " %5, %5, %8\n" I " %6, %6, %8\n" I " %7, %7, %8\n"
#define REP8_1(I) I " %0, %0\n" I " %1, %1\n" I " %2, %2\n" I " %3, %3\n" \
                  I " %4, %4\n" I " %5, %5\n" I " %6, %6\n" I " %7, %7\n"
#define REP8_DPP(I, CTRL) I " %0, %1 " CTRL "\n" I " %1, %2 " CTRL "\n" I " %2, %3 " CTRL "\n" I " %3, %4 " CTRL "\n" \
                          I " %4, %5 " CTRL "\n" I " %5, %6 " CTRL "\n" I " %6, %7 " CTRL "\n" I " %7, %0 " CTRL "\n"

BENCH_F64_2(k_add_f64, "v_add_f64")
BENCH_F64_2(k_mul_f64, "v_mul_f64")
BENCH_F64_3(k_fma_f64, "v_fma_f64")
BENCH_F64_1(k_rndne_f64, "v_rndne_f64")
BENCH_F64_1(k_trunc_f64, "v_trunc_f64")
BENCH_F64_1(k_rcp_f64, "v_rcp_f64")
BENCH_F64_1(k_rsq_f64, "v_rsq_f64")
BENCH_F64_1(k_mov_b64, "v_mov_b64")
BENCH_F64_3(k_pk_fma_f32, "v_pk_fma_f32")
BENCH_F64_2(k_pk_mul_f32, "v_pk_mul_f32")
BENCH_B32_RAW(k_fma_f32, REP8_3("v_fma_f32"))
BENCH_B32_RAW(k_add_f32, REP8_2("v_add_f32"))
BENCH_B32_RAW(k_rcp_f32, REP8_1("v_rcp_f32"))
BENCH_B32_RAW(k_rndne_f32, REP8_1("v_rndne_f32"))
BENCH_B32_RAW(k_mov_b32, REP8_1("v_mov_b32"))
BENCH_B32_RAW(k_cndmask_b32, REP8_2("v_cndmask_b32"))
BENCH_B32_RAW(k_dpp_wave_ror, REP8_DPP("v_mov_b32_dpp", "wave_ror:1 row_mask:0xf bank_mask:0xf"))
BENCH_B32_RAW(k_dpp_row_ror, REP8_DPP("v_mov_b32_dpp", "row_ror:1 row_mask:0xf bank_mask:0xf"))
BENCH_B32_RAW(k_dpp_row_bcast31, REP8_DPP("v_mov_b32_dpp", "row_bcast:31 row_mask:0xf bank_mask:0xf"))
BENCH_B32_RAW(k_bpermute, "ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                          "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n")
BENCH_B32_RAW(k_swap_b32, "v_swap_b32 %0, %1\n v_swap_b32 %2, %3\n v_swap_b32 %4, %5\n v_swap_b32 %6, %7\n"
                          "v_swap_b32 %1, %2\n v_swap_b32 %3, %4\n v_swap_b32 %5, %6\n v_swap_b32 %7, %0\n")

// cmp f64 -> vcc, 8 per iteration
__global__ void k_cmp_f64(double *out, long long *cyc) {
    double a0 = threadIdx.x * 1e-3 + 1.0, b = 1.03;
    unsigned long long acc = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i) {
        asm volatile("v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %1, %2\n"
                     "v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %1, %2\n"
                     "s_or_b64 %0, %0, vcc\n" : "+s"(acc) : "v"(a0), "v"(b) : "vcc", "scc");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(acc & 1);
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
// conversions
__global__ void k_cvt_f32_f64_real(double *out, long long *cyc) {
    double d0 = threadIdx.x * 1e-3 + 1.0, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
    float f0 = 0, f1 = 0, f2 = 0, f3 = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i) {
        asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                     "v_cvt_f64_f32 %4, %0\n v_cvt_f64_f32 %5, %1\n v_cvt_f64_f32 %6, %2\n v_cvt_f64_f32 %7, %3\n"
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + f0 + f1 + f2 + f3;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
// dependent chain latency: fma f64
__global__ void k_fma_f64_dep(double *out, long long *cyc) {
    double a0 = threadIdx.x * 1e-3 + 1.0, b = 1.0000000001, c = 1e-9;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i) {
        asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     : "+v"(a0) : "v"(b), "v"(c));
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

typedef void (*kern_t)(double *, long long *);
struct Entry { const char *name; kern_t k; };

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const char *only = argc > 1 ? argv[1] : nullptr;
    Entry tab[] = {
        {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64", k_fma_f64},
        {"v_fma_f64 (dependent chain)", k_fma_f64_dep},
        {"v_rndne_f64", k_rndne_f64}, {"v_trunc_f64", k_trunc_f64}, {"v_rcp_f64", k_rcp_f64}, {"v_rsq_f64", k_rsq_f64},
        {"v_cmp_lt_f64", k_cmp_f64}, {"v_mov_b64", k_mov_b64},
        {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_mul_f32", k_pk_mul_f32},
        {"v_fma_f32", k_fma_f32}, {"v_add_f32", k_add_f32}, {"v_rcp_f32", k_rcp_f32}, {"v_rndne_f32", k_rndne_f32},
        {"v_mov_b32", k_mov_b32}, {"v_cndmask_b32", k_cndmask_b32},
        {"v_mov_b32_dpp wave_ror:1", k_dpp_wave_ror}, {"v_mov_b32_dpp row_ror:1", k_dpp_row_ror},
        {"v_mov_b32_dpp row_bcast:31", k_dpp_row_bcast31},
        {"ds_bpermute_b32 (x8 + wait)", k_bpermute}, {"v_swap_b32", k_swap_b32},
        {"v_cvt_f32_f64 + v_cvt_f64_f32", k_cvt_f32_f64_real},
    };
    const int blocks = 256;
    double *out; long long *cyc;
    CK(hipMalloc(&out, sizeof(double) * blocks * 1024));
    CK(hipMalloc(&cyc, sizeof(long long) * blocks * 16));
    std::vector<long long> h(blocks * 16);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-34s %10s %10s %10s   (shader cycles per wave-instruction; waves/SIMD = 1,2,4)   eff.clock GHz\n", "instruction", "w=1", "w=2", "w=4");
    for (auto &t : tab) {
        if (only && !strstr(t.name, only)) continue;
        printf("%-34s", t.name);
        double ghz = 0;
        for (int wps : {1, 2, 4}) {
            const int threads = 256 * wps;
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(threads), 0, 0, out, cyc);   // warm
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(threads), 0, 0, out, cyc);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 4 * wps, hipMemcpyDeviceToHost));
            std::vector<long long> v(h.begin(), h.begin() + blocks * 4 * wps);
            std::sort(v.begin(), v.end());
            const double med = (double)v[v.size() / 2];
            // per wave-instruction, normalised per SIMD: cycles * (1 / (8*ITER)) / wps -> issue slots per instr
            printf(" %10.2f", med / (8.0 * ITER) / wps);
            ghz = med / (ms * 1e-3) / 1e9;
        }
        printf("   %.2f\n", ghz);
    }
    return 0;
}
