// ubench_exec.hip -- does an fp64 VALU instruction get cheaper when whole 16-lane rows of the wave are
// masked off in EXEC?  (If the DP pipe skipped inactive rows, a lane mapping that makes the cutoff test
// uniform per 16-lane row would pay without any wave-uniform branch.)  Measurement tool, not product code.
// Each wave issues ITER x 16 independent v_fma_f64 under an EXEC mask selected by `pattern`:
//   0: all 64 lanes   1: lanes 0..31   2: lanes 0..15   3: one lane per row (0,16,32,48)   4: lanes 0..15 + 32..47
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 8192;

__global__ void k(double *out, int pattern)
{
    const int lane = threadIdx.x & 63;
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = lane * 1e-3 + 1.0 + 0.1 * i;
    const double b = 1.0000000001, c = 1e-9;
    bool on = true;
    if (pattern == 1) on = lane < 32;
    if (pattern == 2) on = lane < 16;
    if (pattern == 3) on = (lane & 15) == 0;
    if (pattern == 4) on = (lane & 16) == 0;
    if (on) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    double *out;
    const int blocks = 256 * 4, threads = 256;             // 4 waves per SIMD on 256 CUs
    CK(hipMalloc(&out, sizeof(double) * blocks * threads));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"all 64 lanes", "lanes 0..31", "lanes 0..15", "lanes 0,16,32,48", "rows 0 and 2"};
    printf("%-22s %12s\n", "EXEC pattern", "ns per fma per wave (4 waves/SIMD, all CUs)");
    for (int p = 0; p < 5; ++p) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, p);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, p);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-22s %12.3f\n", names[p], ms * 1e6 / (16.0 * ITER));
    }
    return 0;
}
