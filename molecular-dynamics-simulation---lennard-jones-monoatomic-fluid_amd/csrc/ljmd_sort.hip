// ljmd_sort.hip -- spatial re-ordering of the owned shard (performance only).
//
// Particles are ordered by the Morton (Z-order) code of their cell on a grid of ~1.2 sigma
// cells, so that 64 consecutive slots (one tile = one wave) form a compact blob.  Compact
// tiles are what lets tile_mask_kernel prove that whole 64 x 64 tile pairs lie outside the
// cutoff.  Correctness never depends on the order: bounding boxes are recomputed exactly
// from the actual coordinates before every force evaluation.  The (key, slot) radix sort
// is stable and deterministic, so the summation order -- and hence every bit of the result
// -- is reproducible run to run.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "ljmd_internal.h"

namespace ljmdk {

__device__ __forceinline__ unsigned spread10(unsigned v)   // abcdefghij -> a00b00c00...
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(kBlock) void sort_keys_kernel(SortArgs a)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.P) return;
    unsigned key = 0xffffffffu;                       // padding stays at the end
    if (i < a.S) {
        const double x = a.r[i], y = a.r[a.P + i], z = a.r[2 * (size_t)a.P + i];
        key = 0xfffffffeu;                            // non-finite coordinates: after all cells
        if (isfinite(x) && isfinite(y) && isfinite(z)) {
            const double s = a.ncell / a.L;
            const double fx = (x - a.L * floor(x / a.L)) * s;
            const double fy = (y - a.L * floor(y / a.L)) * s;
            const double fz = (z - a.L * floor(z / a.L)) * s;
            const unsigned cx = min((unsigned)max((int)fx, 0), (unsigned)(a.ncell - 1));
            const unsigned cy = min((unsigned)max((int)fy, 0), (unsigned)(a.ncell - 1));
            const unsigned cz = min((unsigned)max((int)fz, 0), (unsigned)(a.ncell - 1));
            key = (spread10(cx) << 2) | (spread10(cy) << 1) | spread10(cz);
        }
    }
    a.keys[i] = key;
    a.idx[i] = i;
}

__global__ __launch_bounds__(kBlock) void gather3_kernel(const double *src, double *dst, const int *idx, int P)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    const int s = idx[i];
    dst[i] = src[s];
    dst[P + i] = src[P + s];
    dst[2 * (size_t)P + i] = src[2 * (size_t)P + s];
}

__global__ __launch_bounds__(kBlock) void gather_perm_kernel(const int *src, int *dst, const int *idx, int P)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < P) dst[i] = src[idx[i]];
}

// ---- k-d ordering: recursive median split, exactly 64 particles per leaf ---------------
// Level l splits every current segment (a run of whole tiles) in two halves of whole tiles along
// axis l % 3: sorting each segment on that coordinate puts the lower half first.  After
// ceil(log2(#tiles)) levels every tile is a near-cubic box holding exactly 64 particles -- the
// most compact tiles a wave-sized group can have, which is what maximises the number of tile
// pairs the mask can prove to be outside the cutoff.
// key = (segment index << 24) | coordinate quantised to 24 bits: ONE ordinary radix sort per level
// orders every segment along the axis at once.  (A segmented sort of raw doubles gives the same
// order but is ~20x slower in rocPRIM for thousands of small segments.)
constexpr int kCoordBits = 24;

__global__ __launch_bounds__(kBlock) void kd_keys_kernel(const double *coord, const int *idx, unsigned long long *keys,
                                                         const int *seg_offsets, int nseg, int S, double scale)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= S) return;
    int lo = 0, hi = nseg;                              // largest j with seg_offsets[j] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seg_offsets[mid] <= i) lo = mid; else hi = mid;
    }
    const double x = coord[idx[i]] * scale;             // scale = 2^24 / L
    const unsigned long long q = (unsigned long long)fmin(fmax(x, 0.0), (double)((1u << kCoordBits) - 1));
    keys[i] = ((unsigned long long)lo << kCoordBits) | q;
}

__global__ __launch_bounds__(kBlock) void iota_kernel(int *idx, int P)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < P) idx[i] = i;
}

size_t kd_temp_bytes(int count)
{
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned long long *)nullptr,
                                             (unsigned long long *)nullptr, (const int *)nullptr, (int *)nullptr,
                                             count, 0, 64, nullptr);
    return bytes;
}

hipError_t launch_iota(int *idx, int P, hipStream_t s)
{
    hipLaunchKernelGGL(iota_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, idx, P);
    return hipGetLastError();
}

hipError_t kd_level(void *temp, size_t temp_bytes, const double *coord_axis, double L, unsigned long long *keys,
                    unsigned long long *keys_out, int *idx, int *idx_out, int S, int nseg, const int *seg_offsets,
                    hipStream_t s)
{
    hipLaunchKernelGGL(kd_keys_kernel, dim3((S + kBlock - 1) / kBlock), dim3(kBlock), 0, s, coord_axis, idx, keys,
                       seg_offsets, nseg, S, (double)(1u << kCoordBits) / L);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int segbits = 1;
    while ((1 << segbits) < nseg) ++segbits;
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys_out, idx, idx_out, S, 0,
                                              kCoordBits + segbits, s);
}

size_t sort_temp_bytes(int count)
{
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned *)nullptr, (unsigned *)nullptr,
                                             (const int *)nullptr, (int *)nullptr, count, 0, 32, nullptr);
    return bytes;
}

hipError_t launch_sort_keys(const SortArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(sort_keys_kernel, dim3((a.P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t sort_pairs(void *temp, size_t temp_bytes, const unsigned *keys_in, unsigned *keys_out,
                      const int *idx_in, int *idx_out, int count, hipStream_t s)
{
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, count, 0, 32, s);
}

hipError_t launch_gather3(const double *src, double *dst, const int *idx, int P, hipStream_t s)
{
    hipLaunchKernelGGL(gather3_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, src, dst, idx, P);
    return hipGetLastError();
}

hipError_t launch_gather_perm(const int *src, int *dst, const int *idx, int P, hipStream_t s)
{
    hipLaunchKernelGGL(gather_perm_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, src, dst, idx, P);
    return hipGetLastError();
}

}  // namespace ljmdk
