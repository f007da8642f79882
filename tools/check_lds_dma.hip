// check_lds_dma.hip -- where does global_load_lds_dwordx4 put lane l's 16 bytes?  (expected: LDS base + 16 l, so that
// one instruction with lanes l and l + 32 reading the same 512-byte tile axis parks the axis twice in a row).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double *src, double *out)
{
    __shared__ double buf[128];
    const int lane = threadIdx.x;
    buf[lane] = -1.0;
    buf[64 + lane] = -1.0;
    __syncthreads();
    __builtin_amdgcn_global_load_lds(src + (lane & 31) * 2, buf, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[lane] = buf[lane];
    out[64 + lane] = buf[64 + lane];
}
int main()
{
    double h[64], o[128], *d, *dout;
    for (int i = 0; i < 64; ++i) h[i] = 100.0 + i;
    hipMalloc(&d, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int i = 0; i < 128; ++i) ok = ok && o[i] == 100.0 + (i & 63);
    printf("tile parked twice in a row by one global_load_lds_dwordx4: %s\n", ok ? "yes" : "NO");
    if (!ok) for (int i = 0; i < 128; i += 8) printf("%3d: %g %g %g %g %g %g %g %g\n", i, o[i], o[i+1], o[i+2], o[i+3], o[i+4], o[i+5], o[i+6], o[i+7]);
    return ok ? 0 : 1;
}
