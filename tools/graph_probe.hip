// Does hipGraph replay shorten a chain of small DEPENDENT kernels on this stack?  The MD step of a small system is five
// kernels of 4-19 us each on one stream (MEASUREMENTS.md 3.4).  This probe launches chains of five dependent kernels of ~5 us
// (each block spins on s_memtime-free arithmetic over a small array) 2000 times: plain stream launches against one
// captured graph of 20 chains replayed 100 times.  Measurement tool: hipcc --offload-arch=gfx950 -O3 -o graph_probe graph_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void small_kernel(double *a, int n, int iters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = a[i];
    for (int k = 0; k < iters; ++k) v = v * 1.0000001 + 1e-9;       // dependent chain: sets the kernel's duration
    a[i] = v;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 4096, iters = argc > 2 ? std::atoi(argv[2]) : 400;
    const int chains = 2000, per_graph = 20, k_per_chain = 5;
    double *d;
    CK(hipMalloc(&d, n * sizeof(double)));
    CK(hipMemset(d, 0, n * sizeof(double)));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto chain = [&]() { for (int k = 0; k < k_per_chain; ++k) hipLaunchKernelGGL(small_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d, n, iters); };
    for (int w = 0; w < 50; ++w) chain();
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int c = 0; c < chains; ++c) chain();
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    auto t2 = std::chrono::steady_clock::now();
    const double us_cpu = std::chrono::duration<double, std::micro>(t1 - t0).count() / chains;
    const double us_all = std::chrono::duration<double, std::micro>(t2 - t0).count() / chains;
    std::printf("stream launches: %.2f us of CPU per chain of %d kernels, %.2f us per chain until done\n", us_cpu, k_per_chain, us_all);

    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int c = 0; c < per_graph; ++c) chain();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int c = 0; c < chains / per_graph; ++c) CK(hipGraphLaunch(ge, s));
    t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    t2 = std::chrono::steady_clock::now();
    std::printf("graph replay   : %.2f us of CPU per chain, %.2f us per chain until done (graph of %d chains)\n",
                std::chrono::duration<double, std::micro>(t1 - t0).count() / chains,
                std::chrono::duration<double, std::micro>(t2 - t0).count() / chains, per_graph);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipFree(d));
    return 0;
}
