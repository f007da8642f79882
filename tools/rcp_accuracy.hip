// Measures the relative error of v_rcp_f64 and of the refinement variants used by the pair kernels
// against the correctly rounded quotient (measurement tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *e0, double *e1, double *e2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    const double exact = 1.0 / v;
    const double y0 = __builtin_amdgcn_rcp(v);
    double e = fma(-v, y0, 1.0);
    const double yn = fma(y0, e, y0);                 // one Newton step
    const double t = fma(e, e, e);
    const double yh = fma(y0, t, y0);                 // one Halley step
    e0[i] = fabs(y0 - exact) / exact;
    e1[i] = (yn - exact) / exact;
    e2[i] = (yh - exact) / exact;
}
int main()
{
    const int n = 1 << 24;
    std::vector<double> h(n);
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = 0.5 + 1200.0 * ((s >> 11) * (1.0 / 9007199254740992.0)); }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    std::vector<double> a(n), b(n), c(n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0, b1 = 0, b2 = 0; long nz1 = 0, nz2 = 0;
    for (int i = 0; i < n; ++i) { m0 = fmax(m0, a[i]); m1 = fmax(m1, fabs(b[i])); m2 = fmax(m2, fabs(c[i])); b1 += b[i]; b2 += c[i]; nz1 += b[i] != 0; nz2 += c[i] != 0; }
    printf("v_rcp_f64 max rel err      %.3e  (2^%.1f)\n", m0, log2(m0));
    printf("+1 Newton  max rel err     %.3e  mean signed %.3e  not correctly rounded: %.3f%%\n", m1, b1 / n, 100.0 * nz1 / n);
    printf("+1 Halley  max rel err     %.3e  mean signed %.3e  not correctly rounded: %.3f%%\n", m2, b2 / n, 100.0 * nz2 / n);
    return 0;
}
