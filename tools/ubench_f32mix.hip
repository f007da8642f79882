// ubench_f32mix.hip -- wall-clock cost of the instruction kinds the fp32 far-pair loop is made of, per SIMD, at 1..8
// waves per SIMD on the whole chip (256 CUs x 4 SIMDs).  ns per wave-instruction per SIMD = time x 1024 SIMDs x waves
// / instructions; x clock (GHz) = cycles.  Measurement tool, not product code.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_f32mix tools/ubench_f32mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 2048;

#define KERNEL(NAME, NINSTR, BODY)                                                                        \
__global__ __launch_bounds__(64) void NAME(float *out) {                                                  \
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + .1f, a2 = a0 + .2f, a3 = a0 + .3f, a4 = a0 + .4f,     \
          a5 = a0 + .5f, a6 = a0 + .6f, a7 = a0 + .7f, b = 1.0000001f, c = 1e-6f;                         \
    for (int i = 0; i < ITER; ++i)                                                                        \
        asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21");                                       \
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                            \
}                                                                                                         \
static const int NAME##_n = NINSTR;

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define ADD(i) "v_add_f32 %" #i ", %" #i ", %9\n"
#define RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define RND(i) "v_rndne_f32 %" #i ", %" #i "\n"
#define MED(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define MAXF(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define DPP(i) "v_mov_b32_dpp %" #i ", %" #i " wave_ror:1 row_mask:0xf bank_mask:0xf\n"
#define DPPR(i) "v_mov_b32_dpp %" #i ", %" #i " row_ror:1 row_mask:0xf bank_mask:0xf\n"
#define MOV(i) "v_mov_b32 %" #i ", %8\n"
KERNEL(k_fma, 8, R8(FMA))
KERNEL(k_mul, 8, R8(MUL))
KERNEL(k_add, 8, R8(ADD))
KERNEL(k_rcp, 8, R8(RCP))
KERNEL(k_rndne, 8, R8(RND))
KERNEL(k_med3, 8, R8(MED))
KERNEL(k_max, 8, R8(MAXF))
KERNEL(k_cmp, 8, R8(CMP))
KERNEL(k_cndmask, 8, R8(CND))
KERNEL(k_dpp_wave_ror, 8, R8(DPP))
KERNEL(k_dpp_row_ror, 8, R8(DPPR))
KERNEL(k_mov, 8, R8(MOV))
// packed: 64-bit register pairs (doubles as containers)
#define KERNEL_PK(NAME, INSTR)                                                                            \
__global__ __launch_bounds__(64) void NAME(float *out) {                                                  \
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + .1, a2 = a0 + .2, a3 = a0 + .3, a4 = a0 + .4, a5 = a0 + .5, a6 = a0 + .6, a7 = a0 + .7, b = 1.0000001, c = 1e-9; \
    for (int i = 0; i < ITER; ++i)                                                                        \
        asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)              \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
    out[blockIdx.x * 64 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                  \
}                                                                                                         \
static const int NAME##_n = 8;
#define PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %9\n"
KERNEL_PK(k_pk_fma, PKFMA)
KERNEL_PK(k_pk_mul, PKMUL)
KERNEL_PK(k_pk_add, PKADD)
// one fma + the exec-mask region around it, as the compiler emits `if (r2 < rc2) {...}`: cmp, saveexec, branch, body, s_or
#define REGION(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_fma_f32 %" #i ", %" #i ", %8, %9\n1:\n s_or_b64 exec, exec, s[20:21]\n"
KERNEL(k_region_cmp_fma, 16, R8(REGION))
// the same without the branch instruction
#define REGION_NB(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n s_and_saveexec_b64 s[20:21], vcc\n v_fma_f32 %" #i ", %" #i ", %8, %9\n s_or_b64 exec, exec, s[20:21]\n"
KERNEL(k_region_nobranch, 16, R8(REGION_NB))
// cmp + cndmask pairs (select form)
#define SELECT(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
KERNEL(k_cmp_cndmask, 16, R8(SELECT))
// fma with a dependent chain on ONE register (latency)
KERNEL(k_fma_chain, 8, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n")
// fp64 fma for the scale (register pairs)
__global__ __launch_bounds__(64) void k_fma_f64(float *out) {
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + .1, a2 = a0 + .2, a3 = a0 + .3, a4 = a0 + .4, a5 = a0 + .5, a6 = a0 + .6, a7 = a0 + .7, b = 1.0000001, c = 1e-9;
    for (int i = 0; i < ITER; ++i)
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    out[blockIdx.x * 64 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
static const int k_fma_f64_n = 8;

struct K { const char *name; void (*fn)(float *); int n; };
#define E(NAME) {#NAME, NAME, NAME##_n}

int main()
{
    const K ks[] = {E(k_fma), E(k_mul), E(k_add), E(k_mov), E(k_rcp), E(k_rndne), E(k_med3), E(k_max), E(k_cmp), E(k_cndmask),
                    E(k_dpp_wave_ror), E(k_dpp_row_ror), E(k_pk_fma), E(k_pk_mul), E(k_pk_add), E(k_region_cmp_fma),
                    E(k_region_nobranch), E(k_cmp_cndmask), E(k_fma_chain), E(k_fma_f64)};
    const int simds = 1024;
    float *out; CK(hipMalloc(&out, sizeof(float) * 64 * simds * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("ns per wave-instruction per SIMD (VALU + SALU instructions of the body counted), waves per SIMD =\n%-22s", "");
    const int ws[] = {1, 2, 4, 5, 8};
    for (int w : ws) printf(" %7d", w);
    printf("\n");
    for (const K &k : ks) {
        printf("%-22s", k.name);
        for (int w : ws) {
            const int blocks = simds * w;              // one-wave workgroups: w per SIMD when the chip is otherwise idle
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, out);
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 5; ++r) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double instr_per_simd = (double)ITER * k.n * w;
            printf(" %7.3f", best * 1e6 / instr_per_simd);
        }
        printf("\n");
    }
    return 0;
}
