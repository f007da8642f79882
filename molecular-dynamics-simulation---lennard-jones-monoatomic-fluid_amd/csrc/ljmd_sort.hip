// ljmd_sort.hip -- spatial re-ordering of the owned shard (performance only) and the ownership migration of multi-GPU runs.
//
// The shard is kept in k-d order (recursive median split, below): 64 consecutive slots (one tile = one wave) are a
// near-cubic box of exactly 64 particles.  Compact tiles are what lets the tile-pair test prove that whole 64 x 64 tile
// pairs lie outside the cutoff.  Correctness never depends on the order: bounding boxes are recomputed exactly from the
// actual coordinates before every force evaluation.  The (key, slot) radix sorts are stable and deterministic, so the
// summation order -- and hence every bit of the result -- is reproducible run to run.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "ljmd_internal.h"

namespace ljmdk {

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

// ---- ownership migration (multi-GPU): the particles of ALL ranks dealt out again by position -----------------------
// Every rank holds all n positions (exchange buffer [G][3][P], real particles in slots 0..S-1 of each block) and, after
// one all-gather, everybody's ru, v, a and particle ids (migration buffer [G][10][P]).  The deal is a k-d split of the
// n particles into G segments of exactly S = n / G: the same composite-key radix sort as above, over storage indices
// g * P + s, computed redundantly and identically (stable sort, same input bits) by every rank; rank r then takes
// segment r.  Blocks instead of slabs: a rank's surface, and with it the number of column tiles its rows meet at the
// cutoff, does not grow with G.
__global__ __launch_bounds__(kBlock) void iota_blocked_kernel(int *idx, int n, int S, int P)
{
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k < n) idx[k] = (k / S) * P + k % S;
}

__global__ __launch_bounds__(kBlock) void kd_keys_blocked_kernel(const double *pos, int axis, int P, const int *idx,
                                                                 unsigned long long *keys, const int *seg_offsets, int nseg,
                                                                 int n, double scale)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = nseg;                              // largest j with seg_offsets[j] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seg_offsets[mid] <= i) lo = mid; else hi = mid;
    }
    const int st = idx[i], g = st / P, sl = st - g * P;
    const double x = pos[(size_t)g * 3 * P + (size_t)axis * P + sl] * scale;
    const unsigned long long q = (unsigned long long)fmin(fmax(x, 0.0), (double)((1u << kCoordBits) - 1));
    keys[i] = ((unsigned long long)lo << kCoordBits) | q;
}

// own block of the migration buffer: ru, v, a and the particle id of every slot (id < 0: padding)
__global__ __launch_bounds__(kBlock) void migrate_pack_kernel(const double *ru, const double *v, const double *a,
                                                              const int *perm, const int *gid0, double *block, int S, int P)
{
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= P) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        block[(size_t)k * P + s] = ru[(size_t)k * P + s];
        block[(size_t)(3 + k) * P + s] = v[(size_t)k * P + s];
        block[(size_t)(6 + k) * P + s] = a[(size_t)k * P + s];
    }
    const int o = perm[s];
    block[(size_t)9 * P + s] = (s < S && o < S) ? (double)gid0[o] : -1.0;     // ids < 2^31: exact in a double
}

// the rank's new members, in the order of its segment of the deal: positions into a scratch block, ru, v, a and the ids
// into place (their sources are in the migration buffer); padding as after ljmd_set_state
__global__ __launch_bounds__(kBlock) void migrate_select_kernel(const double *pos_all, const double *mig_all, const int *mine,
                                                                double *new_pos, double *ru, double *v, double *a, int *gid0,
                                                                int S, int P)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= P) return;
    if (j >= S) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            new_pos[(size_t)k * P + j] = __builtin_nan("");
            ru[(size_t)k * P + j] = __builtin_nan("");
            v[(size_t)k * P + j] = 0.0;
            a[(size_t)k * P + j] = 0.0;
        }
        gid0[j] = -1;
        return;
    }
    const int st = mine[j], g = st / P, sl = st - g * P;
    const double *pb = pos_all + (size_t)g * 3 * P + sl;
    const double *mb = mig_all + (size_t)g * 10 * P + sl;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        new_pos[(size_t)k * P + j] = pb[(size_t)k * P];
        ru[(size_t)k * P + j] = mb[(size_t)k * P];
        v[(size_t)k * P + j] = mb[(size_t)(3 + k) * P];
        a[(size_t)k * P + j] = mb[(size_t)(6 + k) * P];
    }
    gid0[j] = (int)mb[(size_t)9 * P];
}

__global__ __launch_bounds__(kBlock) void iota_offset_kernel(int *idx, int count, int P, int offset)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < P) idx[i] = i < count ? offset + i : -1;
}

hipError_t launch_iota_blocked(int *idx, int n, int S, int P, hipStream_t s)
{
    hipLaunchKernelGGL(iota_blocked_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, idx, n, S, P);
    return hipGetLastError();
}

hipError_t launch_iota_offset(int *idx, int count, int P, int offset, hipStream_t s)
{
    hipLaunchKernelGGL(iota_offset_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, idx, count, P, offset);
    return hipGetLastError();
}

hipError_t kd_level_blocked(void *temp, size_t temp_bytes, const double *pos, int axis, int P, double L,
                            unsigned long long *keys, unsigned long long *keys_out, int *idx, int *idx_out, int n, int nseg,
                            const int *seg_offsets, hipStream_t s)
{
    hipLaunchKernelGGL(kd_keys_blocked_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pos, axis, P, idx, keys,
                       seg_offsets, nseg, n, (double)(1u << kCoordBits) / L);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int segbits = 1;
    while ((1 << segbits) < nseg) ++segbits;
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys_out, idx, idx_out, n, 0, kCoordBits + segbits, s);
}

hipError_t launch_migrate_pack(const double *ru, const double *v, const double *a, const int *perm, const int *gid0,
                               double *block, int S, int P, hipStream_t s)
{
    hipLaunchKernelGGL(migrate_pack_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, ru, v, a, perm, gid0, block,
                       S, P);
    return hipGetLastError();
}

hipError_t launch_migrate_select(const double *pos_all, const double *mig_all, const int *mine, double *new_pos, double *ru,
                                 double *v, double *a, int *gid0, int S, int P, hipStream_t s)
{
    hipLaunchKernelGGL(migrate_select_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pos_all, mig_all, mine,
                       new_pos, ru, v, a, gid0, S, P);
    return hipGetLastError();
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
