// ljmd_kernels.hip -- gfx950 (MI355X / CDNA4) device kernels of the LJ hot path.
//
// Built with -ffp-contract=off: every fused multiply-add in this file is an explicit
// fma(); everything else keeps the reference's separate roundings.
//
// Reference arithmetic being replaced (paths relative to the reference repo):
//   pair loop            scripts/physics/lj_potential_energy.f90:109-183
//   minimum image        scripts/physics/geometry_pbc.f90:80-88
//   wrap                 scripts/physics/geometry_pbc.f90:39-59
//   velocity Verlet      scripts/physics/verlet.f90:58-95
//   unwrapped update     scripts/md_simulation_program.f90:339-353
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ljmd_internal.h"

namespace ljmdk {

// ---------------------------------------------------------------------------
// wave / block reductions with a FIXED combination order (bitwise reproducible
// run to run: no atomics anywhere in this file).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}

template <int NVAL>
__device__ __forceinline__ void block_sum(double (&v)[NVAL], double *lds /* [NVAL*kWavesPerBlock] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NVAL; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) lds[k * kWavesPerBlock + wave] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NVAL; ++k) {
            double s = lds[k * kWavesPerBlock];
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) s += lds[k * kWavesPerBlock + w];
            v[k] = s;
        }
    }
}

// ---------------------------------------------------------------------------
// One ordered pair (i <- j).  Accumulates the raw (prefactor-free) sums
//   a_i += (2 u^6 - u^3) u * d          (= -dU_r * d * inv_r2, lj_potential_energy.f90:143-155)
//   s12 += u^6 ,  s6 += u^3             (epot, d_epot, dd_epot are linear in these two)
// with u = 1/r^2, d = minimum-image displacement.
// FAST_MIC: d - L*rndne(d/L) with the product fused.  Valid, and bit-identical to
// `d - L*dnint(d*invL)` for every pair that passes r^2 < rc^2, when |d/L| < 2.5
// (L*n exact for |n| <= 2) and rc <= (1-1e-9) L/2 (a rounding tie of d/L means
// |d_mic| ~ L/2 > rc).  The host selects it only under those conditions.
// ---------------------------------------------------------------------------
template <bool FAST_MIC>
__device__ __forceinline__ double mic(double d, double L, double invL)
{
    if constexpr (FAST_MIC) {
        return fma(-L, __builtin_rint(d * invL), d);
    } else {
        return d - L * __builtin_round(d * invL);
    }
}

template <bool FAST_MIC>
__device__ __forceinline__ void pair_accum(double xi, double yi, double zi,
                                           double xj, double yj, double zj,
                                           double L, double invL, double rc2,
                                           double &ax, double &ay, double &az,
                                           double &s12, double &s6)
{
    const double dx = mic<FAST_MIC>(xi - xj, L, invL);
    const double dy = mic<FAST_MIC>(yi - yj, L, invL);
    const double dz = mic<FAST_MIC>(zi - zj, L, invL);
    const double r2 = dx * dx + dy * dy + dz * dz;       // (dx^2 + dy^2) + dz^2 as :129
    if (r2 < rc2) {                                      // strict <, NaN (padding) never passes
        const double u = 1.0 / r2;                       // IEEE divide as :135
        const double u3 = u * u * u;
        const double u6 = u3 * u3;
        s12 += u6;
        s6 += u3;
        const double g = fma(2.0, u6, -u3) * u;
        ax = fma(g, dx, ax);
        ay = fma(g, dy, ay);
        az = fma(g, dz, az);
    }
}

// ---------------------------------------------------------------------------
// K2 (v1): full-matrix gather.  grid = (row tiles, j chunks); one thread per row i.
// The j coordinates are wave-uniform, so they are fetched with scalar loads and fed
// to the VALU as SGPR operands -- no LDS staging, no per-lane address arithmetic.
// pos = exchange buffer in shard-blocked SoA (ljmd.h); rows are the owned shard.
// Output: slab[chunk][axis][row] raw partial accelerations, wg_part[wg][2] = s12,s6.
// ---------------------------------------------------------------------------
template <bool FAST_MIC>
__global__ __launch_bounds__(kBlock) void pair_rows_kernel(PairArgs a)
{
    __shared__ double red[2 * kWavesPerBlock];
    const int row = blockIdx.x * kBlock + threadIdx.x;         // local row in the shard
    const int gi = a.row0 + row;                               // global particle index
    const bool live = row < a.rows;
    const double *own = a.pos + (size_t)a.rank * 3 * a.shard;
    const double nan = __builtin_nan("");
    const double xi = live ? own[row] : nan;
    const double yi = live ? own[a.shard + row] : nan;
    const double zi = live ? own[2 * (size_t)a.shard + row] : nan;

    double ax = 0.0, ay = 0.0, az = 0.0, s12 = 0.0, s6 = 0.0;

    const int j0 = blockIdx.y * a.chunk;
    const int j1 = min(j0 + a.chunk, a.n);
    // rows of this block as a global index range, for the self-pair exclusion
    const int blo = a.row0 + blockIdx.x * kBlock, bhi = blo + kBlock;

    for (int j = j0; j < j1;) {
        const int g = j / a.shard;                              // source shard block (uniform)
        const int jl = j - g * a.shard;
        const int jend_blk = min(j1, (g + 1) * a.shard);
        const double *bx = a.pos + (size_t)g * 3 * a.shard;
        const double *by = bx + a.shard, *bz = by + a.shard;
        const int cnt = jend_blk - j;
        // split at the block's own rows so that only that segment pays the j != i test
        int k = 0;
        while (k < cnt) {
            const int jg = j + k;
            if (jg >= blo && jg < bhi) {
                const int stop = min(cnt, bhi - j);
                for (; k < stop; ++k) {
                    if (j + k != gi)
                        pair_accum<FAST_MIC>(xi, yi, zi, bx[jl + k], by[jl + k], bz[jl + k],
                                             a.L, a.invL, a.rc2, ax, ay, az, s12, s6);
                }
            } else {
                const int stop = (jg < blo) ? min(cnt, blo - j) : cnt;
#pragma unroll 4
                for (; k < stop; ++k)
                    pair_accum<FAST_MIC>(xi, yi, zi, bx[jl + k], by[jl + k], bz[jl + k],
                                         a.L, a.invL, a.rc2, ax, ay, az, s12, s6);
            }
        }
        j = jend_blk;
    }

    if (live) {
        double *s = a.slab + (size_t)blockIdx.y * 3 * a.shard;
        s[row] = ax;
        s[a.shard + row] = ay;
        s[2 * (size_t)a.shard + row] = az;
    }
    double v[2] = {s12, s6};
    block_sum<2>(v, red);
    if (threadIdx.x == 0) {
        double *w = a.wg_part + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        w[0] = v[0];
        w[1] = v[1];
    }
}

// ---------------------------------------------------------------------------
// K1: drift + wrap + first half-kick + unwrapped update, one thread per particle.
//   r  = (r + v*dt) + a*dt_square_half        verlet.f90:58-60   (left-to-right, unfused)
//   r  = r - L*floor(r*invL)                  geometry_pbc.f90:54-56
//   v  = v + a*dt_half                        verlet.f90:72-74
//   ru = ru + mic(r_new - r_old)              md_simulation_program.f90:341-351 (dnint)
// HBM-bound: reads r,v,a,ru (96 B) + writes r,v,ru (72 B) = 168 B per particle.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void drift_kick_kernel(IntegrateArgs a)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.rows) return;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const size_t o = (size_t)ax * a.shard + i;
        const double r0 = a.r[o], v0 = a.v[o], acc = a.a[o];
        double r1 = (r0 + v0 * a.dt) + acc * a.dt_sq_half;
        r1 = r1 - a.L * __builtin_floor(r1 * a.invL);
        double d = r1 - r0;
        d = d - a.L * __builtin_round(d * a.invL);
        a.r[o] = r1;
        a.v[o] = v0 + acc * a.dt_half;
        a.ru[o] = a.ru[o] + d;
    }
}

// ---------------------------------------------------------------------------
// K3: reduce the partial-acceleration slabs in fixed chunk order, restore the x24
// prefactor (lj_potential_energy.f90:189-191), optionally apply the second half-kick
// (verlet.f90:86-88) and emit per-block partial sums of vx^2, vy^2, vz^2
// (verlet.f90:93-95 keeps the three sums separate).
// ---------------------------------------------------------------------------
template <bool KICK>
__global__ __launch_bounds__(kBlock) void reduce_kick_kernel(IntegrateArgs a)
{
    __shared__ double red[3 * kWavesPerBlock];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    double k2[3] = {0.0, 0.0, 0.0};
    if (i < a.rows) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const size_t o = (size_t)ax * a.shard + i;
            double s = a.slab[o];
            for (int c = 1; c < a.nslab; ++c) s += a.slab[(size_t)c * 3 * a.shard + o];
            const double acc = 24.0 * s;
            a.a[o] = acc;
            if constexpr (KICK) {
                const double v1 = a.v[o] + acc * a.dt_half;
                a.v[o] = v1;
                k2[ax] = v1 * v1;
            }
        }
    }
    if constexpr (KICK) {
        block_sum<3>(k2, red);
        if (threadIdx.x == 0) {
            double *w = a.ke_part + 3 * (size_t)blockIdx.x;
            w[0] = k2[0];
            w[1] = k2[1];
            w[2] = k2[2];
        }
    }
}

// Kinetic-energy partials only: per block sum of (vx^2 + vy^2 + vz^2), the fused form
// of md_simulation_program.f90:238-240.
__global__ __launch_bounds__(kBlock) void kinetic_fused_kernel(IntegrateArgs a)
{
    __shared__ double red[kWavesPerBlock];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    double k[1] = {0.0};
    if (i < a.rows) {
        const double vx = a.v[i], vy = a.v[a.shard + i], vz = a.v[2 * (size_t)a.shard + i];
        k[0] = vx * vx + vy * vy + vz * vz;
    }
    block_sum<1>(k, red);
    if (threadIdx.x == 0) {
        double *w = a.ke_part + 3 * (size_t)blockIdx.x;
        w[0] = k[0];
        w[1] = 0.0;
        w[2] = 0.0;
    }
}

// ---------------------------------------------------------------------------
// K4: one block folds the per-workgroup partials into ONE partial record of this
// rank for this step, appended to the scalar ring at *ring_pos:
//   rec = { S12, S6, Kx, Ky, Kz, 0, 0, 0 }    (kPartialStride doubles)
// The host (ljmd_combine_scalars) adds the ranks in rank order and applies the
// prefactors and tail constants.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void finalize_kernel(FinalizeArgs a)
{
    __shared__ double red[5 * kWavesPerBlock];
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int w = threadIdx.x; w < a.n_wg; w += kBlock) {
        v[0] += a.wg_part[2 * (size_t)w];
        v[1] += a.wg_part[2 * (size_t)w + 1];
    }
    for (int b = threadIdx.x; b < a.n_ke; b += kBlock) {
        v[2] += a.ke_part[3 * (size_t)b];
        v[3] += a.ke_part[3 * (size_t)b + 1];
        v[4] += a.ke_part[3 * (size_t)b + 2];
    }
    block_sum<5>(v, red);
    if (threadIdx.x == 0) {
        const unsigned pos = *a.ring_pos;
        double *rec = a.ring + (size_t)(pos % a.ring_cap) * kPartialStride;
        rec[0] = v[0];
        rec[1] = v[1];
        rec[2] = v[2];
        rec[3] = v[3];
        rec[4] = v[4];
        rec[5] = 0.0;
        rec[6] = 0.0;
        rec[7] = 0.0;
        *a.ring_pos = pos + 1;
    }
}

// ---------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------
hipError_t launch_pair_rows(const PairArgs &a, bool fast_mic, dim3 grid, hipStream_t s)
{
    if (fast_mic)
        hipLaunchKernelGGL(pair_rows_kernel<true>, grid, dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL(pair_rows_kernel<false>, grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_drift_kick(const IntegrateArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(drift_kick_kernel, dim3((a.rows + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_reduce_kick(const IntegrateArgs &a, bool kick, hipStream_t s)
{
    const dim3 grid((a.rows + kBlock - 1) / kBlock);
    if (kick)
        hipLaunchKernelGGL(reduce_kick_kernel<true>, grid, dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL(reduce_kick_kernel<false>, grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_kinetic_fused(const IntegrateArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(kinetic_fused_kernel, dim3((a.rows + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_finalize(const FinalizeArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace ljmdk
