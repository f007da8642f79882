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
//
// Layout conventions: ljmd_internal.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ljmd_internal.h"

namespace ljmdk {

// ---------------------------------------------------------------------------
// wave / block reductions with a FIXED combination order: every floating-point sum in this file
// is bitwise reproducible run to run.  The only atomics are integer ones that cannot change a result:
// the blocks-done ticket of kick_finalize_kernel and the integer histogram of rdf_histogram_kernel.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}

__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));  // fmin ignores NaN
    return v;
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
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

// ===========================================================================
// GENERIC pair kernel: exact for ANY finite input (positions far outside the box,
// rc arbitrarily close to L/2).  Per pair it performs the reference's operations with
// the reference's roundings -- dnint minimum image, (dx^2+dy^2)+dz^2, IEEE divide --
// except that it accumulates the linear sums s12 = sum u^6, s6 = sum u^3 instead of
// epot / d_epot / dd_epot separately.  Full matrix (each ordered pair), one thread per
// row, j broadcast through scalar loads.  Used when the fast path's preconditions do
// not hold (ljmd_capi.cpp: fast_path_ok).
// ===========================================================================
__global__ __launch_bounds__(kBlock) void pair_rows_generic_kernel(PairArgs a)
{
    __shared__ double red[2 * kWavesPerBlock];
    const int row = blockIdx.x * kBlock + threadIdx.x;         // slot in the owned block, < P
    const double *own = a.pos + (size_t)a.rank * 3 * a.P;
    const double xi = own[row], yi = own[a.P + row], zi = own[2 * (size_t)a.P + row];   // NaN on padding

    double ax = 0.0, ay = 0.0, az = 0.0, s12 = 0.0, s6 = 0.0;

    const int j0 = blockIdx.y * a.chunk;                       // index into the n real particles,
    const int j1 = min(j0 + a.chunk, a.n);                     // rank-major: j = g * S + slot
    for (int j = j0; j < j1;) {
        const int g = j / a.S;
        const int jl = j - g * a.S;
        const int jend = min(j1, (g + 1) * a.S);
        const double *bx = a.pos + (size_t)g * 3 * a.P;
        const double *by = bx + a.P, *bz = by + a.P;
        const int cnt = jend - j;
        const bool own_block = (g == a.rank);
        for (int k = 0; k < cnt; ++k) {
            const double dx0 = xi - bx[jl + k], dy0 = yi - by[jl + k], dz0 = zi - bz[jl + k];
            const double dx = dx0 - a.L * __builtin_round(dx0 * a.invL);      // geometry_pbc.f90:86
            const double dy = dy0 - a.L * __builtin_round(dy0 * a.invL);
            const double dz = dz0 - a.L * __builtin_round(dz0 * a.invL);
            const double r2 = dx * dx + dy * dy + dz * dz;                    // :129
            if (r2 < a.rc2 && !(own_block && jl + k == row)) {                // :132 + self exclusion
                const double u = 1.0 / r2;                                    // :135
                const double u3 = u * u * u;                                  // :136
                const double u6 = u3 * u3;                                    // :137
                s12 += u6;
                s6 += u3;
                const double mdu = 2.0 * u6 - u3;                             // = -dU_r, :143
                ax += mdu * dx * u;                                           // :148-155
                ay += mdu * dy * u;
                az += mdu * dz * u;
            }
        }
        j = jend;
    }

    double *s = a.slab + (size_t)blockIdx.y * 3 * a.P;
    s[row] = ax;
    s[a.P + row] = ay;
    s[2 * (size_t)a.P + row] = az;
    double v[2] = {s12, s6};
    block_sum<2>(v, red);
    if (threadIdx.x == 0) {
        double *w = a.wg_part + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        w[0] = v[0];
        w[1] = v[1];
    }
}

// ===========================================================================
// FAST path building blocks.  Preconditions (checked on the host):
//   (a) every coordinate lies within a span < 2.4 L (true after any wrap), so |d/L| < 2.5,
//       n = rndne(d/L) has |n| <= 2 and L*n is exact: fma(-L, n, d) == d - L*n rounded once,
//       i.e. the same value the reference computes;
//   (b) rc <= (1 - 1e-9) L/2: rndne and dnint differ only on exact ties of d/L, where
//       |d_mic| ~ L/2 > rc, so the pair fails r^2 < rc^2 either way.
// Differences from the reference per pair, all <= ~1 ulp of the term: r^2 and the force
// use fma contraction, 1/r^2 is v_rcp_f64 + one Halley step instead of the IEEE divide.
// ===========================================================================
__device__ __forceinline__ double mic_fast(double d, double L, double invL)
{
    return fma(-L, __builtin_rint(d * invL), d);
}

__device__ __forceinline__ double rcp_newton(double x)
{
    // v_rcp_f64 delivers ~24-26 good bits; ONE cubically convergent (Halley) step takes the relative
    // error e to e^3 (< 2^-70): y = y0 (1 + e + e^2), e = 1 - x y0.  3 fma instead of the 4 of two
    // Newton steps; result within 1 ulp of the IEEE quotient (tests/test_gpu_parity.py).
    const double y0 = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y0, 1.0);
    const double t = fma(e, e, e);
    return fma(y0, t, y0);
}

template <bool EXCLUDE_SELF>
__device__ __forceinline__ void pair_fast(double xi, double yi, double zi,
                                          double xj, double yj, double zj,
                                          double L, double invL, double rc2, bool is_self,
                                          double &ax, double &ay, double &az,
                                          double &s12, double &s6)
{
    const double dx = mic_fast(xi - xj, L, invL);
    const double dy = mic_fast(yi - yj, L, invL);
    const double dz = mic_fast(zi - zj, L, invL);
    const double r2 = fma(dz, dz, fma(dy, dy, dx * dx));
    bool in = r2 < rc2;                                  // strict <; NaN (padding) never passes
    if constexpr (EXCLUDE_SELF) in = in && !is_self;
    if (in) {
        const double u = rcp_newton(r2);
        const double u3 = u * u * u;
        const double u6 = u3 * u3;
        s12 += u6;
        s6 += u3;
        const double g = fma(2.0, u6, -u3) * u;          // = -dU_r * inv_r2
        ax = fma(g, dx, ax);
        ay = fma(g, dy, ay);
        az = fma(g, dz, az);
    }
}

// Lower bound of |d - m L| over d in [lo, hi], m integer, |d| < 2.5 L.
__device__ __forceinline__ double axis_gap(double lo, double hi, double L)
{
    double g = __builtin_inf();
#pragma unroll
    for (int m = -2; m <= 2; ++m) {
        const double c = m * L;
        if (lo <= c && c <= hi) return 0.0;
        g = fmin(g, fmin(fabs(lo - c), fabs(hi - c)));
    }
    return g;
}

// The tile-pair test of the geometry pre-pass (tile_mask_kernel, tile_class): squared lower bound of the minimum-image
// distance between the boxes of tiles I and J.  A pair of tiles is skipped only when this exceeds rc^2 (1 + 1e-10).
__device__ __forceinline__ double tile_gap2(const double *bbox, int I, int J, double L)
{
    const double *bi = bbox + (size_t)I * kBoxStride;
    const double *bj = bbox + (size_t)J * kBoxStride;
    double d2 = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double g = axis_gap(bi[k] - bj[3 + k], bi[3 + k] - bj[k], L);
        d2 += g * g;
    }
    return d2;
}

// ---------------------------------------------------------------------------
// K2 (gather, tiled): one wave per 64-particle row tile I (lane = particle), 4 waves per
// workgroup.  The wave walks its row of the tile-pair mask (bit J set = tile J holds at
// least one particle that can be within rc of tile I -- built by tile_mask_kernel from exact
// bounding boxes) and for every set bit evaluates the 64 x 64 ordered pairs: the j
// coordinates are wave-uniform, fetched with scalar loads and used as SGPR operands.
// grid = (TB / 4, column-tile slices).  Output as in the generic kernel.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pair_tiles_kernel(PairArgs a)
{
    __shared__ double red[2 * kWavesPerBlock];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Il = blockIdx.x * kWavesPerBlock + wave;          // own tile, wave-uniform
    const int I = a.rank * a.TB + Il;
    const int row = Il * kTile + lane;
    const double *own = a.pos + (size_t)a.rank * 3 * a.P;
    const double xi = own[row], yi = own[a.P + row], zi = own[2 * (size_t)a.P + row];

    double ax = 0.0, ay = 0.0, az = 0.0, s12 = 0.0, s6 = 0.0;

    const uint64_t *mrow = a.mask + (size_t)Il * a.W;
    // this workgroup's slice of the column tiles: [J0, J1)  (a.chunk tiles; small systems get many slices)
    const int J0 = blockIdx.y * a.chunk, J1 = min(J0 + a.chunk, a.T);
    for (int w = J0 >> 6; w <= (J1 - 1) >> 6 && J0 < J1; ++w) {
        uint64_t m;
        if (a.inline_mask) {                                 // lane = column tile of this word: tile_mask_kernel's test
            const int Jl = w * 64 + lane;
            const bool keep = Jl < a.T && (!(tile_gap2(a.bbox, I, Jl, a.L) > a.rc2_skin) || Jl == I);
            m = __ballot(keep);
        } else {
            m = mrow[w];
        }
        const int lo = max(J0 - w * 64, 0), hi = min(J1 - w * 64, 64);      // bits of this word inside the slice
        m &= (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            const int J = w * 64 + b;
            const int gj = (a.G == 1) ? 0 : J / a.TB;
            const int jl = (J - gj * a.TB) * kTile;
            const double *bx = a.pos + (size_t)gj * 3 * a.P + jl;
            const double *by = bx + a.P, *bz = by + a.P;
            // 8 j at a time: 3 x 64-byte scalar loads up front, then 8 pair evaluations on SGPR operands
            if (J == I) {
                for (int j0 = 0; j0 < kTile; j0 += 8) {
                    double xj[8], yj[8], zj[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { xj[k] = bx[j0 + k]; yj[k] = by[j0 + k]; zj[k] = bz[j0 + k]; }
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        pair_fast<true>(xi, yi, zi, xj[k], yj[k], zj[k], a.L, a.invL, a.rc2, j0 + k == lane,
                                        ax, ay, az, s12, s6);
                }
            } else {
                for (int j0 = 0; j0 < kTile; j0 += 8) {
                    double xj[8], yj[8], zj[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { xj[k] = bx[j0 + k]; yj[k] = by[j0 + k]; zj[k] = bz[j0 + k]; }
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        pair_fast<false>(xi, yi, zi, xj[k], yj[k], zj[k], a.L, a.invL, a.rc2, false,
                                         ax, ay, az, s12, s6);
                }
            }
        }
    }

    double *s = a.slab + (size_t)blockIdx.y * 3 * a.P;
    s[row] = ax;
    s[a.P + row] = ay;
    s[2 * (size_t)a.P + row] = az;
    double v[2] = {s12, s6};
    block_sum<2>(v, red);
    if (threadIdx.x == 0) {
        double *w = a.wg_part + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        w[0] = v[0];
        w[1] = v[1];
    }
}

// ===========================================================================
// K2 (Newton-3): every unordered pair exactly once.
//
// One wave = one ROW GROUP of 4 tiles (256 particles, 4 per lane, positions and partial
// accelerations in registers for the whole kernel).  For each COLUMN TILE (64 particles, one
// per lane) the wave performs 64 steps: evaluate up to 4 pairs (one per row particle of the
// lane) against the column particle currently held by the lane, add +f to the row side and
// -f to the column side (lj_potential_energy.f90:153-159), then rotate the column particle
// (position + partial acceleration, 12 dwords) one lane around the wave with DPP wave_ror.
// After 64 steps every column particle is back in its home lane carrying the sum of the
// forces from all 256 row particles; that block goes to slab_j[row group][column slot]
// (one writer per block: no atomics, fixed summation order, reduced by reduce_kick_kernel).
//
// Ownership: row group A evaluates column group B = A + d (mod NG) for d = 0 .. Dmax = NG/2
// (d = NG/2 only from the lower index), so each unordered pair of groups appears once; inside
// the diagonal group (d = 0) tile pairs k < l are taken once and a tile against itself uses
// steps 1..31 plus the lower half of step 32.  Tile pairs proven outside the cutoff by the
// mask are skipped (all four row tiles out => the column tile is not even loaded).
// grid = (ceil(NG / 4), offset chunks); the waves of a workgroup are independent.
// ===========================================================================

// UNIFORM: the host-side geometry proved that every pair of this (row group, column tile) takes the
// SAME periodic image on each axis, n = (sx, sy, sz) / L with |n| <= 2.  Then
//   d = (xi - xj) - s      (two roundings: the difference, then the exact-product subtraction)
// is bit-identical to fma(-L, rndne((xi - xj) * invL), xi - xj) and costs 2 instead of 4
// instructions per axis.
__device__ __forceinline__ double dpp_rotate(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x13C /* wave_ror:1 */, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x13C, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// STRADDLE form of the minimum image.  The differences xi - xj of a (row group, column tile) pass cover a range narrower
// than L (tile frames), so when no single image serves all pairs the range contains exactly ONE half-integer multiple of
// L, (n + 1/2) L or (n - 1/2) L, and every pair takes image n or its neighbour on that side.  The pass folds n L + L/2
// (or n L - L/2) into the column tile once (n3_tile_pass), so that e = xi - xj'' changes sign exactly where the image
// does, and
//     d = e - copysign(L/2, e)
// is the minimum-image displacement: one fp64 subtraction and a v_bfi_b32 on the sign word instead of the multiply,
// v_rndne_f64 and fma of d - L rndne(d / L) -- 2 fp64 instructions per axis instead of 4.  On the tie e = 0 both images
// are L/2 away, beyond any rc the fast path accepts; the roundings of the two folds move the decision point by ~ulp(2L),
// eight orders of magnitude inside the 1e-9 L/2 margin between rc and L/2.  half = 0 makes it the plain difference.
__device__ __forceinline__ double straddle(double e, double half)
{
    return e - __builtin_copysign(half, e);
}

// the displacement of one pair in the loop variant NU (see pair_n3); sx, sy, sz = the half-box constants of the
// straddle forms (L/2 on a straddling axis, 0 on a plain one)
template <int NU>
__device__ __forceinline__ void pair_disp(double xi, double yi, double zi, double xj, double yj, double zj, double L,
                                          double invL, double sx, double sy, double sz, double &dx, double &dy, double &dz)
{
    static_assert(NU == 7 || NU == 8 || NU == 32 || NU == 33 || NU == 34 || NU == 40, "pair_disp: loop variant");
    dx = xi - xj; dy = yi - yj; dz = zi - zj;
    if constexpr (NU == 7) {
        dx = fma(-L, __builtin_rint(dx * invL), dx);
        dy = fma(-L, __builtin_rint(dy * invL), dy);
        dz = fma(-L, __builtin_rint(dz * invL), dz);
    }
    if constexpr (NU == 32 || NU == 40) dx = straddle(dx, sx);
    if constexpr (NU == 33 || NU == 40) dy = straddle(dy, sy);
    if constexpr (NU == 34 || NU == 40) dz = straddle(dz, sz);
}

// ENERGY = false: the two energy sums are not accumulated (steps whose observables nobody reads, see N3Args::energy)
template <bool LANE_PRED, int NU, bool INNER = false, bool ENERGY = true>
__device__ __forceinline__ void pair_n3(double xi, double yi, double zi,
                                        double xj, double yj, double zj,
                                        double L, double invL, double rc2, bool lane_ok,
                                        double sx, double sy, double sz,
                                        double &ax, double &ay, double &az,
                                        double &jx, double &jy, double &jz,
                                        double &s12, double &s6)
{
    // NU == 7: all three axes with the general minimum image d - L rndne(d / L) (4 instructions per axis)
    // NU == 8: the common image of every axis is folded into the column tile (n3_tile_pass): d = xi - xj
    // NU == 32 + a: axis a STRADDLES one half-box distance (below), the two others as NU == 8
    // NU == 40: the straddle form on all three axes, the half-box constant of a plain axis being 0
    // (the per-pair shift-subtracting forms NU = 0..6, 16..18 and the one-general-axis forms 24..26 are what the fold
    //  and the straddle form replaced)
    double dx, dy, dz;
    pair_disp<NU>(xi, yi, zi, xj, yj, zj, L, invL, sx, sy, sz, dx, dy, dz);
    const double r2 = fma(dz, dz, fma(dy, dy, dx * dx));
    bool in = true;
    if constexpr (!INNER) in = r2 < rc2;       // INNER: the boxes prove r^2 < rc^2 for every pair
    if constexpr (LANE_PRED) in = in && lane_ok;
    if (in) {
        const double u = rcp_newton(r2);
        const double u3 = u * u * u;
        const double u6 = u3 * u3;
        if constexpr (ENERGY) {
            s12 += u6;
            s6 += u3;
        }
        const double g = fma(2.0, u6, -u3) * u;
        ax = fma(g, dx, ax);
        ay = fma(g, dy, ay);
        az = fma(g, dz, az);
        jx = fma(-g, dx, jx);
        jy = fma(-g, dy, jy);
        jz = fma(-g, dz, jz);
    }
}

// pair_n3 in two halves, for the batched reciprocal of column_tile_loop: the displacement and r^2 ...
template <int NU>
__device__ __forceinline__ void pair_geom(double xi, double yi, double zi, double xj, double yj, double zj, double L,
                                          double invL, double sx, double sy, double sz, double &dx, double &dy,
                                          double &dz, double &r2)
{
    pair_disp<NU>(xi, yi, zi, xj, yj, zj, L, invL, sx, sy, sz, dx, dy, dz);
    r2 = fma(dz, dz, fma(dy, dy, dx * dx));
}

// ... and everything behind the reciprocal u = 1 / r^2
template <bool INNER, bool ENERGY>
__device__ __forceinline__ void pair_apply(double u, double dx, double dy, double dz, double r2, double rc2, double &ax,
                                           double &ay, double &az, double &jx, double &jy, double &jz, double &s12,
                                           double &s6)
{
    bool in = true;
    if constexpr (!INNER) in = r2 < rc2;
    if (in) {
        const double u3 = u * u * u;
        const double u6 = u3 * u3;
        if constexpr (ENERGY) {
            s12 += u6;
            s6 += u3;
        }
        const double g = fma(2.0, u6, -u3) * u;
        ax = fma(g, dx, ax);
        ay = fma(g, dy, ay);
        az = fma(g, dz, az);
        jx = fma(-g, dx, jx);
        jy = fma(-g, dy, jy);
        jz = fma(-g, dz, jz);
    }
}

// 64 rotation steps of one column tile against the wave's 4 row tiles (see pair_n3_kernel).
// The column POSITIONS are read-only, so they do not have to travel through DPP: the tile is parked in LDS
// twice in a row (entries i and i + 64), and at step s lane l reads entry (l + 64 - s) -- the particle that a
// rotation by s lanes would have brought to it -- with an immediate offset and no address arithmetic: three
// ds_read_b64 on the LDS port, prefetched one step ahead, instead of six DPP moves on the VALU.  Only the partial
// accelerations still rotate.
#ifndef LJMD_N3_UNROLL
#define LJMD_N3_UNROLL 8
#endif
#ifndef LJMD_BATCH_RCP
#define LJMD_BATCH_RCP 1       // one reciprocal per step for all row tiles of a lane (column_tile_loop<..., BATCH>)
#endif
constexpr int kLdsAxis = 2 * kTile;         // doubles per axis in the parked column tile

// BATCH (all RT row tiles active, no padding slot on either side): ONE reciprocal serves the RT pair evaluations of a
// step.  v_rcp_f64 is a quarter-rate instruction (16 cycles against 4 for an fp64 FMA), so it and its Halley step are
// a quarter of a pair evaluation; with p = r0^2 r1^2 r2^2 r3^2, y = 1/p (v_rcp_f64 + one Halley step, correctly rounded
// as before) the four reciprocals are  1/r0^2 = r1^2 (y r2^2 r3^2), 1/r1^2 = r0^2 (y r2^2 r3^2), ... : 9 multiplications
// and one reciprocal chain instead of four chains, 384 instead of 424 SIMD cycles per step.  Every r^2 of a full tile
// pair is finite and far from under/overflow (r^2 <= 3 (L/2 + tile)^2, product of four <= 1e17 at L = 69), lanes
// outside the cutoff compute a reciprocal they do not use.  Each 1/r^2 carries three roundings instead of one
// (<= 1.5 ulp); the parity tests hold their bounds unchanged.
template <int RT, int NU, bool MASKED, bool INNER, bool BATCH = false, bool ENERGY = true>
__device__ __forceinline__ void column_tile_loop(const double (&xi)[RT], const double (&yi)[RT],
                                                 const double (&zi)[RT], double (&ax)[RT],
                                                 double (&ay)[RT], double (&az)[RT],
                                                 const double *park, int ns, unsigned mb,
                                                 double L, double invL, double rc2, double sx, double sy, double sz,
                                                 double &jx, double &jy, double &jz, double &s12, double &s6)
{
    static_assert(!BATCH || (!MASKED && (RT == 2 || RT == 4)), "batched reciprocal: all row tiles, 2 or 4 of them");
    static_assert(kTile / 2 % LJMD_N3_UNROLL == 0, "half a pass is a whole number of unrolled bodies");
    // The rotation steps s0 .. s0 + ns - 1 of the pass (a whole pass: 0 .. 63; one side of a tie worked from both sides,
    // N3Args::both_ties: 0 .. 31 or 1 .. 32; ns is a multiple of the unroll count).  park = &lds[lane - s0]: entry (lane + 64 - s0 - s) is the
    // particle that step s0 + s brings to the lane (s0 + s = 0: the lane's own).
    double xj, yj, zj;
    double nx = park[kTile], ny = park[kLdsAxis + kTile], nz = park[2 * kLdsAxis + kTile];
    for (int sb = 0; sb < ns; sb += LJMD_N3_UNROLL, park -= LJMD_N3_UNROLL) {
#pragma unroll
    for (int s = 0; s < LJMD_N3_UNROLL; ++s) {
        xj = nx; yj = ny; zj = nz;
        nx = park[kTile - 1 - s];                       // next step's particle; the last prefetch of a part is unused
        ny = park[kLdsAxis + kTile - 1 - s];
        nz = park[2 * kLdsAxis + kTile - 1 - s];
        if constexpr (BATCH) {
            double dx[RT], dy[RT], dz[RT], r2[RT], u[RT];
#pragma unroll
            for (int k = 0; k < RT; ++k)
                pair_geom<NU>(xi[k], yi[k], zi[k], xj, yj, zj, L, invL, sx, sy, sz, dx[k], dy[k], dz[k], r2[k]);
            if constexpr (RT == 4) {
                const double pab = r2[0] * r2[1], pcd = r2[2] * r2[3];
                const double y = rcp_newton(pab * pcd);
                const double rab = y * pcd, rcd = y * pab;
                u[0] = r2[1] * rab; u[1] = r2[0] * rab; u[2] = r2[3] * rcd; u[3] = r2[2] * rcd;
            } else {
                const double y = rcp_newton(r2[0] * r2[1]);
                u[0] = r2[1] * y; u[1] = r2[0] * y;
            }
#pragma unroll
            for (int k = 0; k < RT; ++k)
                pair_apply<INNER, ENERGY>(u[k], dx[k], dy[k], dz[k], r2[k], rc2, ax[k], ay[k], az[k], jx, jy, jz, s12, s6);
        } else {
#pragma unroll
            for (int k = 0; k < RT; ++k)
                if (!MASKED || ((mb >> k) & 1u))
                    pair_n3<false, NU, INNER, ENERGY>(xi[k], yi[k], zi[k], xj, yj, zj, L, invL, rc2, true, sx, sy, sz,
                                              ax[k], ay[k], az[k], jx, jy, jz, s12, s6);
        }
        jx = dpp_rotate(jx); jy = dpp_rotate(jy); jz = dpp_rotate(jz);
    }
    }
}

// Wave-uniform image classification of one axis: raw differences xi - xj of all pairs lie in
// [lo, hi]; returns true and the shift n*L when rndne(d * invL) is the same n for every d in it.
__device__ __forceinline__ bool uniform_image(double lo, double hi, double L, double invL, double &shift)
{
    const double tlo = lo * invL, thi = hi * invL;
    const double n = __builtin_rint(0.5 * (tlo + thi));
    shift = n * L;                                       // exact for |n| <= 2
    return (tlo > n - 0.5 + 1e-9) && (thi < n + 0.5 - 1e-9) && (fabs(n) <= 2.0);
}

#ifdef LJMD_VARIANT_STATS
__device__ unsigned long long g_variant_stats[80];
#endif
#ifdef LJMD_WAVE_TRACE
// measurement build (tools/wave_trace.py): per wave of the small-system pair kernels 8 words -- s_memrealtime (100 MHz) at
// the start, after the pass descriptor, after the rotation loop(s) and at the end, HW_ID, XCC_ID, (heavy passes, work item)
constexpr int kTraceWaves = 1 << 16;
__device__ unsigned long long g_wave_trace[kTraceWaves * 8];
__device__ __forceinline__ unsigned long long trace_now() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void trace_store(unsigned wave_index, unsigned long long t0, unsigned long long t1, unsigned long long t2,
                                            unsigned heavy, unsigned item)
{
    if (wave_index >= (unsigned)kTraceWaves) return;
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
    unsigned long long *o = g_wave_trace + (size_t)wave_index * 8;
    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = trace_now(); o[4] = hw; o[5] = xcc; o[6] = heavy; o[7] = item;
}
#endif

// W waves per workgroup (W = 1, 2, 4): the W waves hold W CONSECUTIVE row groups and walk the same column tiles in
// lock step -- wave w sees column group A0 + e at its own offset d = e - w -- so that their column-side partial
// accelerations can be added through LDS (fixed order w = 0 .. W-1) and leave the chip as ONE slab_j block per
// (workgroup, column tile) instead of one per (row group, column tile): W times less column-side slab memory and
// traffic, and the column tile is fetched once per workgroup's L1 instead of once per wave.  One s_barrier per
// column tile; comb[] is double-buffered so that wave 0's combine overlaps the other waves' next tile.
template <int W>
__device__ __forceinline__ void wave_lds_sync()
{
    if constexpr (W == 1) {
        __syncthreads();                                   // one wave per workgroup: orders this wave's own LDS traffic
    } else {
        // the parked tile is private to the wave and LDS executes a wave's instructions in order: only the
        // compiler has to be kept from moving the reads across the writes
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ===========================================================================
// Column CLUSTERS for the passes at the cutoff boundary.
//
// A 64 x 64 tile pair that straddles the cutoff sphere is evaluated in full by the rotation loop although about 45 % of
// its pairs lie outside (every lane meets every column particle: a rotation step holds all sub-boxes of the row tile at
// once, so nothing inside a pass can be skipped).  With rc = 0.49 L the sphere's surface is nearly a plane on the scale
// of a tile: seen from the row group, the column tile's particles are inside or outside according to their PROJECTION
// on the direction n from the row group to the column tile.  A cluster pass therefore
//   * sorts the column tile along n (a 64-key bitonic network across the wave: fp32 projection | lane index),
//   * cuts it into 4 clusters of 16 (slabs ~1 sigma thick instead of a 4.3 sigma box),
//   * skips a (row tile k, cluster) pair when  min_j n.(xj + s) - max_i n.x_i > rc  -- |n| <= 1, so the projection of a
//     displacement never exceeds its length: every pair of it fails the reference's r^2 < rc^2 (lj_potential_energy.f90:132)
//     -- with the row tile's maximum taken from its exact box (support function), precomputed per pass by
//     tile_class_kernel together with n (desc2),
//   * and runs the kept clusters REPLICATED over the four 16-lane DPP rows: every lane meets the cluster's 16 particles
//     in 16 steps (positions parked in LDS twice in a row per cluster, partial accelerations rotating with row_ror:1),
//     after which the four rows' partial accelerations of a column particle are added (fixed order) and stored.
// Measured on the bench configuration (tools/slab_study.py): 62.2 % of all pairs evaluated instead of 69.8 % (49 % are
// inside the cutoff); clusters of 16 k-d leaves instead of projection slabs: 65.8 %.
// Only passes without a general (per-pair) periodic image take this path: there the distance the loop computes is the
// plain difference to the shifted column tile, which is what the projection bounds.
// ===========================================================================
constexpr int kClu = 16;                    // particles per column cluster = lanes per DPP row
constexpr int kNClu = kTile / kClu;

__device__ __forceinline__ double dpp_row_rotate(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x121 /* row_ror:1 */, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x121, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// float -> unsigned with the same order; the low 6 bits make room for the lane index.  Clearing them rounds the value
// DOWN (towards -inf) for either sign, so a decoded key is a lower bound of the projection it was made from.
__device__ __forceinline__ unsigned ordered_key(float f)
{
    const unsigned b = __float_as_uint(f);
    return ((b & 0x80000000u) ? ~b : (b | 0x80000000u)) & ~63u;
}

__device__ __forceinline__ float key_value(unsigned u)
{
    u &= ~63u;
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// value of lane (l ^ J) for every lane l, without an address register: DPP inside a row of 16 where the pattern is a
// quad permutation or a rotation, the LDS crossbar (ds_swizzle, no memory access) for 4 and 16
template <int J>
__device__ __forceinline__ unsigned lane_xor(unsigned v)
{
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16, "lane_xor: 32 goes through v_permlane32_swap");
    if constexpr (J == 1) return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, false);
    if constexpr (J == 2) return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, false);
    if constexpr (J == 8) return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128 /* row_ror:8 */, 0xF, 0xF, false);
    if constexpr (J == 4) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);   // bit mode: and 0x1f, xor 4
    return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);                         // xor 16
}

// lanes that keep the SMALLER key in the compare-exchange (block size K, distance J) of an ascending bitonic sort
constexpr unsigned long long bitonic_min_lanes(int K, int J)
{
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (((l & K) == 0) == ((l & J) == 0)) m |= 1ull << l;
    return m;
}

template <int K, int J>
__device__ __forceinline__ unsigned bitonic_step(unsigned key)
{
    unsigned mn, mx;
    if constexpr (J == 32) {
        // v_permlane32_swap: first result = the low half twice, second = the high half twice
        const auto h = __builtin_amdgcn_permlane32_swap(key, key, false, false);
        mn = min(key, h[1]);                            // what lanes 0..31 keep
        mx = max(key, h[0]);                            // what lanes 32..63 keep
    } else {
        const unsigned other = lane_xor<J>(key);
        mn = min(key, other);
        mx = max(key, other);
    }
    // the lane set is a compile-time constant: a literal SGPR pair as the select mask (computed from the lane index it
    // is 21 hoisted compares, i.e. 42 SGPRs the allocator then spills)
    constexpr unsigned long long lanes = bitonic_min_lanes(K, J);
    unsigned out;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(out) : "v"(mx), "v"(mn), "s"(lanes));
    return out;
}

// ascending bitonic sort of 64 distinct keys, one per lane
__device__ __forceinline__ unsigned wave_sort(unsigned key)
{
    key = bitonic_step<2, 1>(key);
    key = bitonic_step<4, 2>(key); key = bitonic_step<4, 1>(key);
    key = bitonic_step<8, 4>(key); key = bitonic_step<8, 2>(key); key = bitonic_step<8, 1>(key);
    key = bitonic_step<16, 8>(key); key = bitonic_step<16, 4>(key); key = bitonic_step<16, 2>(key);
    key = bitonic_step<16, 1>(key);
    key = bitonic_step<32, 16>(key); key = bitonic_step<32, 8>(key); key = bitonic_step<32, 4>(key);
    key = bitonic_step<32, 2>(key); key = bitonic_step<32, 1>(key);
    key = bitonic_step<64, 32>(key); key = bitonic_step<64, 16>(key); key = bitonic_step<64, 8>(key);
    key = bitonic_step<64, 4>(key); key = bitonic_step<64, 2>(key); key = bitonic_step<64, 1>(key);
    return key;
}

// sum over the four DPP rows (lanes l, l ^ 16, l ^ 32, l ^ 48), the same bits in every lane:
// (r0 + r1) + (r2 + r3) for every row (additions commute)
__device__ __forceinline__ double rows_sum(double v)
{
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    v += __hiloint2double((int)lane_xor<16>(hi), (int)lane_xor<16>(lo));
    lo = (unsigned)__double2loint(v);
    hi = (unsigned)__double2hiint(v);
    const auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
}

// 16 rotation steps of one column cluster (replicated over the four DPP rows) against the wave's RT row tiles;
// park = &parked[32 * cluster + (lane & 15)]: entry + 16 - s is the particle a rotation by s inside the row brings
template <int RT, bool MASKED, bool BATCH, bool ENERGY>
__device__ __forceinline__ void column_cluster_loop(const double (&xi)[RT], const double (&yi)[RT], const double (&zi)[RT],
                                                    double (&ax)[RT], double (&ay)[RT], double (&az)[RT], double xj,
                                                    double yj, double zj, const double *park, unsigned mb, double rc2,
                                                    double &jx, double &jy, double &jz, double &s12, double &s6)
{
    static_assert(!BATCH || (!MASKED && (RT == 2 || RT == 4)), "batched reciprocal: all row tiles, 2 or 4 of them");
    double nx = xj, ny = yj, nz = zj;
#pragma unroll LJMD_N3_UNROLL
    for (int s = 0; s < kClu; ++s) {
        xj = nx; yj = ny; zj = nz;
        nx = park[kClu - 1 - s];                        // next step's particle; the last prefetch is unused
        ny = park[kLdsAxis + kClu - 1 - s];
        nz = park[2 * kLdsAxis + kClu - 1 - s];
        if constexpr (BATCH) {
            double dx[RT], dy[RT], dz[RT], r2[RT], u[RT];
#pragma unroll
            for (int k = 0; k < RT; ++k)
                pair_geom<8>(xi[k], yi[k], zi[k], xj, yj, zj, 0.0, 0.0, 0.0, 0.0, 0.0, dx[k], dy[k], dz[k], r2[k]);
            if constexpr (RT == 4) {
                const double pab = r2[0] * r2[1], pcd = r2[2] * r2[3];
                const double y = rcp_newton(pab * pcd);
                const double rab = y * pcd, rcd = y * pab;
                u[0] = r2[1] * rab; u[1] = r2[0] * rab; u[2] = r2[3] * rcd; u[3] = r2[2] * rcd;
            } else {
                const double y = rcp_newton(r2[0] * r2[1]);
                u[0] = r2[1] * y; u[1] = r2[0] * y;
            }
#pragma unroll
            for (int k = 0; k < RT; ++k)
                pair_apply<false, ENERGY>(u[k], dx[k], dy[k], dz[k], r2[k], rc2, ax[k], ay[k], az[k], jx, jy, jz, s12, s6);
        } else {
#pragma unroll
            for (int k = 0; k < RT; ++k)
                if (!MASKED || ((mb >> k) & 1u))
                    pair_n3<false, 8, false, ENERGY>(xi[k], yi[k], zi[k], xj, yj, zj, 0.0, 0.0, rc2, true, 0.0, 0.0, 0.0,
                                                     ax[k], ay[k], az[k], jx, jy, jz, s12, s6);
        }
        jx = dpp_row_rotate(jx); jy = dpp_row_rotate(jy); jz = dpp_row_rotate(jz);
    }
}

// One column tile against the wave's RT row tiles, cluster by cluster (see above).  Writes the column-side block itself
// (the sort permutes the tile: lane l of DPP row c ends with the sums of the particle whose slot is `slot`).
template <int RT, bool ENERGY>
__device__ __forceinline__ void n3_cluster_pass(const N3Args &a, int lane, int Al, int c, unsigned mb, unsigned desc,
                                                const double (&xi)[RT], const double (&yi)[RT], const double (&zi)[RT],
                                                double (&ax)[RT], double (&ay)[RT], double (&az)[RT], double *parked,
                                                double *out /* slab_j block */, double &s12, double &s6)
{
    const size_t P = a.P;
    const int gj = (a.G == 1) ? 0 : c / a.TB;
    const double *cb = a.pos + (size_t)gj * 3 * P + (size_t)(c - gj * a.TB) * kTile + lane;
    double xj = cb[0], yj = cb[P], zj = cb[2 * P];
    xj += (double)((int)((desc >> 11) & 7u) - 2) * a.L;          // the pass's common image, folded in once
    yj += (double)((int)((desc >> 14) & 7u) - 2) * a.L;
    zj += (double)((int)((desc >> 17) & 7u) - 2) * a.L;

    // direction and per-row-tile thresholds of the pass (tile_class_kernel); wave-uniform -> scalar loads
    const float *d2 = a.desc2 + ((size_t)Al * a.T + c) * 8;
    const float nx = d2[0], ny = d2[1], nz = d2[2];
    float thr[4] = {d2[3], d2[4], d2[5], d2[6]};

    // sort the tile along n: key = projection (fp64 arithmetic on the fp32 direction, rounded once) | lane
    const float proj = (float)fma((double)nz, zj, fma((double)ny, yj, (double)nx * xj));
    const unsigned sorted = wave_sort(ordered_key(proj) | (unsigned)lane);
    const int slot = (int)(sorted & 63u);                          // the particle this lane holds from now on

    // lane l fetches particle `slot` through LDS, then the tile is parked cluster by cluster, each twice in a row
    wave_lds_sync<1>();                                            // the previous pass's reads are done
    parked[lane] = xj;
    parked[kLdsAxis + lane] = yj;
    parked[2 * kLdsAxis + lane] = zj;
    wave_lds_sync<1>();
    xj = parked[slot];
    yj = parked[kLdsAxis + slot];
    zj = parked[2 * kLdsAxis + slot];
    wave_lds_sync<1>();
    const int row = lane >> 4, i = lane & (kClu - 1);
    {
        double *pk = parked + 2 * kClu * row + i;
        pk[0] = xj; pk[kClu] = xj;
        pk[kLdsAxis] = yj; pk[kLdsAxis + kClu] = yj;
        pk[2 * kLdsAxis] = zj; pk[2 * kLdsAxis + kClu] = zj;
    }
    wave_lds_sync<1>();

    // (NOT unrolled: the two 16-step loop bodies below are ~13 KB of code each; one copy of each stays in the instruction
    //  cache across clusters and passes, four copies do not -- measured 18.3 vs 17.1 ms without clusters when unrolled)
#pragma unroll 1
    for (int cl = 0; cl < kNClu; ++cl) {
        // smallest projection in the cluster (sorted: its first lane), rounded down
        const float kmin = key_value((unsigned)__builtin_amdgcn_readlane((int)sorted, cl * kClu));
        unsigned mbc = 0;
#pragma unroll
        for (int k = 0; k < RT; ++k)
            if (((mb >> k) & 1u) && kmin <= thr[k]) mbc |= 1u << k;
        double jx = 0.0, jy = 0.0, jz = 0.0;
        if (mbc) {
            const double *pk = parked + 2 * kClu * cl + i;
            const double x0 = pk[kClu], y0 = pk[kLdsAxis + kClu], z0 = pk[2 * kLdsAxis + kClu];   // particle i of the cluster
            if (mbc == ((1u << RT) - 1u))
                column_cluster_loop<RT, false, (RT == 2 || RT == 4) && LJMD_BATCH_RCP, ENERGY>(
                    xi, yi, zi, ax, ay, az, x0, y0, z0, pk, mbc, a.rc2, jx, jy, jz, s12, s6);
            else
                column_cluster_loop<RT, true, false, ENERGY>(xi, yi, zi, ax, ay, az, x0, y0, z0, pk, mbc, a.rc2, jx, jy,
                                                             jz, s12, s6);
            // the four DPP rows hold four partial sums of every particle of the cluster: add them (same bits in every row)
            jx = rows_sum(jx);
            jy = rows_sum(jy);
            jz = rows_sum(jz);
        }
#ifdef LJMD_VARIANT_STATS
        if (lane == 0) atomicAdd(&g_variant_stats[64 + __builtin_popcount(mbc)], 1ull);   // clusters by active row tiles
#endif
        if (row == cl) {                                           // sorted position lane = 16 cl + i: this lane's particle
            out[slot] = jx;
            out[kTile + slot] = jy;
            out[2 * kTile + slot] = jz;
        }
    }
}

// One column tile against the wave's RT row tiles: the rotation steps s0 .. s0 + ns - 1 (a whole pass: 0 .. 63) in the loop
// variant the pass descriptor names.  The tile's positions are loaded here (one particle per lane) and parked in LDS.
// The column-side sums start at zero and rotate one lane per step: after the last step lane l holds those of column slot
// l - s_end (s_end = the steps of the pass that lie behind it: 64 brings every particle home).
template <int RT, int W, bool ENERGY>
__device__ __forceinline__ void n3_tile_pass(const N3Args &a, int lane, int c, int d, int l, int s0, int ns, unsigned mb,
                                             unsigned desc, const double (&xi)[RT],
                                             const double (&yi)[RT], const double (&zi)[RT], double (&ax)[RT],
                                             double (&ay)[RT], double (&az)[RT], double *parked,
                                             double &jx, double &jy, double &jz, int &s_end, double &s12, double &s6)
{
    const size_t P = a.P;
    const int gj = (a.G == 1) ? 0 : c / a.TB;          // rank block holding the column tile
    const double *cb = a.pos + (size_t)gj * 3 * P + (size_t)(c - gj * a.TB) * kTile + lane;
    double xj = cb[0], yj = cb[P], zj = cb[2 * P];

    // the pass descriptor (tile_class_kernel): loop variant, INNER, FULL and the common image per axis
    const int nu = (int)((desc >> 4) & 31u);
    const bool inner = ((desc >> 9) & 1u) != 0, full = ((desc >> 10) & 1u) != 0;
    const double sx = (double)((int)((desc >> 11) & 7u) - 2) * a.L;     // n L, exact for |n| <= 2
    const double sy = (double)((int)((desc >> 14) & 7u) - 2) * a.L;
    const double sz = (double)((int)((desc >> 17) & 7u) - 2) * a.L;

    // the column tile is one of the wave's own row tiles (tile l against itself): the general minimum image, no fold
    const bool diag = d == 0 && ((mb >> l) & 1u);
    // A common image of the whole tile pair is subtracted from the column tile ONCE, here, instead of from every
    // pair's difference: d = xi - (xj + nL).  Three additions per pass replace one subtraction per pair and imaged
    // axis, and the loop variants with a common image collapse into the one without (nu 16, 17, 18, 0 -> 8).
    // xj + nL is rounded at a magnitude <= 2 L where the reference rounds xi - xj at <= L before its exact
    // minimum-image correction (geometry_pbc.f90:86): the same order of error (<= 2 ulp(L)), not the same bits.
    // An axis without a common image STRADDLES (nu 27, 28, 29: that axis alone; 30: two or three of them): the half-box
    // distance goes into the column tile as well and the loop runs the straddle form (pair_disp).
    const bool general_all = nu == 7 || diag;
    int loop = 7;
    double hx = 0.0, hy = 0.0, hz = 0.0;                 // half-box constants of the straddle forms
    if (!general_all) {
        xj += sx; yj += sy; zj += sz;                   // 0.0 where the image is n = 0
        loop = 8;
        if (nu >= 27 && nu <= 30) {
            const double half = 0.5 * a.L;              // exact
            const unsigned st = (desc >> 22) & 7u, sg = (desc >> 25) & 7u;    // straddling axes; 1 = towards +L/2
            if (st & 1u) { hx = half; xj += (sg & 1u) ? half : -half; }
            if (st & 2u) { hy = half; yj += (sg & 2u) ? half : -half; }
            if (st & 4u) { hz = half; zj += (sg & 4u) ? half : -half; }
            loop = nu == 30 ? 40 : 32 + (nu - 27);
        }
    }
    wave_lds_sync<W>();                                // the previous tile's reads are done
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const double val = q == 0 ? xj : q == 1 ? yj : zj;
        parked[q * kLdsAxis + lane] = val;
        parked[q * kLdsAxis + kTile + lane] = val;
    }
    wave_lds_sync<W>();
    if (diag) {
        // tile l against itself on the steps 1 .. 31 and the lower half of step 32 (each unordered pair once), against the
        // group's earlier row tiles k < l on every step.  With one tile per group nothing lies beyond step 32.
        const int t0 = RT == 1 ? max(s0, 1) : s0, t1 = RT == 1 ? min(s0 + ns, kTile / 2 + 1) : s0 + ns;
        const double *q0 = parked + lane + kTile;       // entry lane + 64 - s
        for (int s = t0; s < t1; ++s) {
            const double cx = q0[-s], cy = q0[kLdsAxis - s], cz = q0[2 * kLdsAxis - s];
#pragma unroll
            for (int k = 0; k < RT; ++k) {
                if (!((mb >> k) & 1u)) continue;
                if (k == l) {
                    if (s >= 1 && s <= kTile / 2)
                        pair_n3<true, 7, false, ENERGY>(xi[k], yi[k], zi[k], cx, cy, cz, a.L, a.invL, a.rc2,
                                         s < kTile / 2 || lane < kTile / 2, 0.0, 0.0, 0.0, ax[k], ay[k], az[k], jx,
                                         jy, jz, s12, s6);
                } else {
                    pair_n3<false, 7, false, ENERGY>(xi[k], yi[k], zi[k], cx, cy, cz, a.L, a.invL, a.rc2, true, 0.0,
                                      0.0, 0.0, ax[k], ay[k], az[k], jx, jy, jz, s12, s6);
                }
            }
            jx = dpp_rotate(jx); jy = dpp_rotate(jy); jz = dpp_rotate(jz);
        }
        s_end = max(t0, t1);
        return;
    }
    s_end = s0 + ns;
#define LJMD_LOOP(NU_, MASKED_, INNER_)                                                                      \
    column_tile_loop<RT, NU_, MASKED_, INNER_, false, ENERGY>(xi, yi, zi, ax, ay, az, parked + lane - s0, ns, mb, a.L, a.invL,  \
                                                  a.rc2, hx, hy, hz, jx, jy, jz, s12, s6)
#define LJMD_LOOP_ALL(NU_, INNER_)                                                                           \
    column_tile_loop<RT, NU_, false, INNER_, (RT == 2 || RT == 4) && LJMD_BATCH_RCP, ENERGY>(                              \
        xi, yi, zi, ax, ay, az, parked + lane - s0, ns, mb, a.L, a.invL, a.rc2, hx, hy, hz, jx, jy, jz, s12, s6)
    // all row tiles active AND no padding slot anywhere in the tile pair: the unmasked loop with the batched
    // reciprocal; otherwise the masked loop (correct for any mb; a NaN padding slot must not enter a product)
    const bool all4 = mb == ((1u << RT) - 1u) && (full || !LJMD_BATCH_RCP);
#ifdef LJMD_VARIANT_STATS
    if (lane == 0)          // measurement build only (tools/variant_stats.py): row tiles evaluated per class
    {
        atomicAdd(&g_variant_stats[(nu & 31) * 2 + (inner ? 1 : 0)], (unsigned long long)__builtin_popcount(mb));
        atomicAdd(&g_variant_stats[56 + __builtin_popcount(mb)], 1ull);      // column-tile passes by active row tiles
    }
#endif
    if (loop == 8 && inner) { if (all4) LJMD_LOOP_ALL(8, true); else LJMD_LOOP(8, true, true); }
    else if (loop == 8)     { if (all4) LJMD_LOOP_ALL(8, false); else LJMD_LOOP(8, true, false); }
    else if (loop == 32)    { if (all4) LJMD_LOOP_ALL(32, false); else LJMD_LOOP(32, true, false); }
    else if (loop == 33)    { if (all4) LJMD_LOOP_ALL(33, false); else LJMD_LOOP(33, true, false); }
    else if (loop == 34)    { if (all4) LJMD_LOOP_ALL(34, false); else LJMD_LOOP(34, true, false); }
    else if (loop == 40)    { if (all4) LJMD_LOOP_ALL(40, false); else LJMD_LOOP(40, true, false); }
    else                    { if (all4) LJMD_LOOP_ALL(7, false); else LJMD_LOOP(7, true, false); }
#undef LJMD_LOOP_ALL
#undef LJMD_LOOP
}

// (defined with the geometry pre-pass below)
template <int RT>
__device__ __forceinline__ bool tile_class(const GeometryArgs &a, double invL, double rc2, int S, int Al, int c,
                                           unsigned &desc_out, unsigned *desc_far, float *desc2);

template <int MIN_WAVES, int RT, int W, bool ENERGY>
__global__ __launch_bounds__(kTile * W, MIN_WAVES) void pair_n3_kernel(N3Args a)
{
    __shared__ double parked_all[W][3 * kLdsAxis];   // per wave: the column tile, twice in a row per axis
    __shared__ double comb[W > 1 ? 2 : 1][W][3][W > 1 ? kTile : 1];
    __shared__ int comb_on[2][W];
    const int lane = threadIdx.x & 63;
    const int wv = W == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // XCD-aware work mapping: the hardware deals workgroups round-robin over the 8 XCDs (b and b + 8 share one, each
    // XCD has its own L2).  With the plain mapping the row groups A, A+1, ... that read the same column tiles (B = A + d
    // for the few d of a slice) sit on 8 different XCDs and every one of them fetches the tile through the fabric.
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if (a.xcd_remap) {                              // host guarantees gridDim.x % (8 * chunk) == 0
        // an XCD takes chunks of `xcd_remap` consecutive row groups, the chunks dealt round-robin over the XCDs: the
        // row groups inside a chunk share their column tiles through the XCD's L2, while every XCD still sees the same
        // statistical mix of heavy and light work items at any time (ONE contiguous eighth per XCD measured 10 %
        // slower: for a given slice the work of neighbouring row groups is correlated, and an XCD that holds only
        // heavy items stalls the round-robin dispatcher for the others)
        const unsigned id = blockIdx.y * gridDim.x + blockIdx.x, per = gridDim.x >> 3, C = (unsigned)a.xcd_remap;
        const unsigned xcd = id & 7u, slot = id >> 3, s = slot % per;
        by = slot / per;
        bx = ((s / C) * 8u + xcd) * C + s % C;
    }
#ifdef LJMD_WAVE_TRACE
    const unsigned long long tr0 = trace_now();
    unsigned long long tr1 = tr0;
    unsigned tr_heavy = 0;
#endif
    const int Al = (int)bx * W + wv;                           // owned row group, wave-uniform
    const bool active = Al < a.NGo;
    const int A = a.rank * a.NGo + Al;                         // its global index
    const int A0 = a.rank * a.NGo + (int)bx * W;               // the workgroup's first row group
    const size_t P = a.P;
    const double *own = a.pos + (size_t)a.rank * 3 * P;

    double xi[RT], yi[RT], zi[RT];
    double ax[RT], ay[RT], az[RT];
    double s12 = 0.0, s6 = 0.0;
#pragma unroll
    for (int k = 0; k < RT; ++k) {
        const size_t slot = (size_t)(active ? RT * Al + k : 0) * kTile + lane;
        xi[k] = own[slot];
        yi[k] = own[P + slot];
        zi[k] = own[2 * P + slot];
        ax[k] = ay[k] = az[k] = 0.0;
    }

    // UNITS of work.  A row group walks the offsets e = 0 .. Dmax (+ W - 1: e is counted from the workgroup's FIRST row
    // group, wave w is at its own offset d = e - w) and at every offset the RT column tiles l of the group there: unit
    // u = e * RT + l, one PASS.  A work item (grid.y slice) takes `uchunk` consecutive units.  Large systems: a few offsets'
    // worth; small and middle-sized ones cut finer -- down to one pass per item -- so that the 1024 SIMDs get several items
    // each (profiles/r04_unit_sweep.txt; cutting the passes themselves into halves or quarters was built, measured there and
    // removed: a part repeats the item's fixed cost).  Every wave of the workgroup runs the same units (one barrier per
    // column tile when W > 1).
    const int u0 = (int)by * a.uchunk;
    const int nt = max(0, min(a.uchunk, (a.Dmax + W) * RT - u0));
    // The tie d = NG / 2 (NG even) is one pass for TWO row groups.  both_ties (one rank, W = 1): instead of one side owning
    // the whole pass and the other idling, A < B takes the steps 0 .. 31 against B and B the steps 1 .. 32 against A.  Step
    // s of (A, B) pairs row i with column i - s; step s' of (B, A) pairs row j with column j - s', i.e. the shift -s' of the
    // first pass: the two halves cover every shift mod 64 exactly once, and the work of the row groups is equal.
    const bool both = W == 1 && a.both_ties != 0;

    // unit t of this work item: its column tile c, offset d, rotation steps s0 .. s0 + ns - 1, slab block, and the mask bits of
    // the wave's RT row tiles (0 = nothing to do: not owned, or every row tile proven outside the cutoff)
    auto tile_of = [&](int t, int &c, int &d, int &l, int &s0, int &ns, size_t &blk, unsigned &desc) -> unsigned {
        const int u = u0 + t;
        const int e = u / RT;
        l = u - e * RT;
        d = e - wv;
        const bool valid = active && d >= 0 && d <= a.Dmax;
        int B = A0 + e;
        if (B >= a.NG) B -= a.NG;
        const bool tie = 2 * d == a.NG;
        const bool work = valid && ((d == 0) || (2 * d < a.NG) || (tie && (A < B || both)));
        c = RT * B + l;                                 // column tile (global)
        blk = (size_t)c * a.CS + (size_t)(a.by_group ? (int)bx : e / W);   // N3Args::slab_j
        s0 = 0;
        ns = kTile;
        if (tie && both) {                              // the lower row group: steps 0 .. 31, the upper one: 1 .. 32
            s0 = A < B ? 0 : 1;
            ns = kTile / 2;
        }
        unsigned mb = 0;
        desc = 0;
        if (work) {
            // wave-uniform by construction: keep it in SGPRs so that the per-row-tile tests are scalar branches
            if constexpr (RT <= 2 && W == 1) {
                // small single-rank systems (N3Args::inline_class): the wave works its pass descriptor out itself
                // -- the same function tile_class_kernel runs -- and the step saves a launch
                if (a.inline_class) {
                    GeometryArgs ga;
                    ga.pos = nullptr; ga.bbox = const_cast<double *>(a.bbox); ga.pos_tc = nullptr; ga.mask = nullptr; ga.mask_far = nullptr;
                    ga.P = a.P; ga.G = a.G; ga.rank = a.rank; ga.TB = a.TB; ga.T = a.T; ga.W = a.W; ga.RT = RT;
                    ga.L = a.L; ga.invL = a.invL; ga.rc2_skin = a.rc2_skin; ga.rsplit2 = 0.0; ga.rvfar2 = 0.0; ga.pertile_images = 0;
                    ga.both_ties = a.both_ties;
                    unsigned dsc = 0;
                    (void)tile_class<RT>(ga, a.invL, a.rc2, a.S, Al, c, dsc, nullptr, nullptr);
                    desc = (unsigned)__builtin_amdgcn_readfirstlane((int)dsc);
                } else {
                    desc = (unsigned)__builtin_amdgcn_readfirstlane((int)a.desc[(size_t)Al * a.T + c]);
                }
            } else {
                desc = (unsigned)__builtin_amdgcn_readfirstlane((int)a.desc[(size_t)Al * a.T + c]);
            }
            mb = desc & 15u;
            if (d == 0) mb &= (2u << l) - 1u;              // diagonal group: row tile k <= column tile l
        }
        return mb;
    };

    {
        double *parked = parked_all[wv];
        int buf = 0;
        for (int t = 0; t < nt; ++t) {
            int c, d, l, s0, ns;
            size_t blk;
            unsigned desc;
            const unsigned mb = tile_of(t, c, d, l, s0, ns, blk, desc);
            const bool have = mb != 0;
            int s_end = kTile;                                 // column slot of lane l's sums after the pass: l - s_end
#ifdef LJMD_WAVE_TRACE
            if (t == 0) tr1 = trace_now();
            tr_heavy += have ? 1u : 0u;
#endif
            double jx = 0.0, jy = 0.0, jz = 0.0;
            bool stored = false;
            const bool pertile = have && ((desc >> 21) & 1u);
            if (pertile) {
                // per-tile periodic images on one axis (tile_class): the lane's row particles of this pass
                const int axis = (int)((desc >> 22) & 3u);
#pragma unroll
                for (int k = 0; k < RT; ++k) {
                    const double sh = (double)((int)((desc >> (24 + 2 * k)) & 3u) - 1) * a.L;
                    if (axis == 0) xi[k] -= sh; else if (axis == 1) yi[k] -= sh; else zi[k] -= sh;
                }
            }
            if (have) {
                // two-level energy sums: a pass (<= 256 terms per lane) sums into its own pair, which is added to the
                // work item's running pair once -- the rounding error of a lane's sum no longer grows with the number of
                // column tiles a work item walks (LJMD_N3_TARGET_WAVES, n), 2 additions per 64 rotation steps
                double p12 = 0.0, p6 = 0.0;
                if constexpr (W == 1 && RT == kRowTiles) {
                    // boundary pass: cluster by cluster (tile_class_kernel) -- whole passes only
                    if (((desc >> 20) & 1u) && ns == kTile) {
                        n3_cluster_pass<RT, ENERGY>(a, lane, Al, c, mb, desc, xi, yi, zi, ax, ay, az, parked,
                                                    a.slab_j + blk * (3 * kTile), p12, p6);
                        stored = true;
                    }
                }
                if (!stored)
                    n3_tile_pass<RT, W, ENERGY>(a, lane, c, d, l, s0, ns, mb, desc, xi, yi, zi, ax, ay, az, parked, jx, jy, jz,
                                                s_end, p12, p6);
                s12 += p12;
                s6 += p6;
            }
            if (pertile) {                                     // the lane's row particles as they are, bit for bit
                const int axis = (int)((desc >> 22) & 3u);
#pragma unroll
                for (int k = 0; k < RT; ++k) {
                    const double x0 = own[(size_t)axis * P + (size_t)(RT * Al + k) * kTile + lane];
                    if (axis == 0) xi[k] = x0; else if (axis == 1) yi[k] = x0; else zi[k] = x0;
                }
            }
            if constexpr (W == 1) {
                if (have && !stored) {
                    double *o = a.slab_j + blk * (3 * kTile) + ((lane - s_end) & (kTile - 1));
                    o[0] = jx;
                    o[kTile] = jy;
                    o[2 * kTile] = jz;
                }
                if (lane == 0) a.flag_j[blk] = have ? 1 : 0;
            } else {
                if (have) {
                    comb[buf][wv][0][lane] = jx;
                    comb[buf][wv][1][lane] = jy;
                    comb[buf][wv][2][lane] = jz;
                }
                if (lane == 0) comb_on[buf][wv] = have ? 1 : 0;
                __syncthreads();
                if (wv == 0) {
                    double tx = 0.0, ty = 0.0, tz = 0.0;       // 0 + x == x: the first live wave's block passes unchanged
                    bool any = false;
#pragma unroll
                    for (int w = 0; w < W; ++w)
                        if (comb_on[buf][w]) {
                            any = true;
                            tx += comb[buf][w][0][lane];
                            ty += comb[buf][w][1][lane];
                            tz += comb[buf][w][2][lane];
                        }
                    if (any) {
                        double *o = a.slab_j + blk * (3 * kTile) + lane;
                        o[0] = tx;
                        o[kTile] = ty;
                        o[2 * kTile] = tz;
                    }
                    if (lane == 0) a.flag_j[blk] = any ? 1 : 0;
                }
                buf ^= 1;
            }
        }
    }

#ifdef LJMD_WAVE_TRACE
    const unsigned long long tr2 = trace_now();
#endif
    if (active) {
        double *si = a.slab_i + (size_t)by * 3 * P;
#pragma unroll
        for (int k = 0; k < RT; ++k) {
            const size_t slot = (size_t)(RT * Al + k) * kTile + lane;
            si[slot] = ax[k];
            si[P + slot] = ay[k];
            si[2 * P + slot] = az[k];
        }
    }
    // without the sums the step's potential-energy observables come out as NaN, never as a stale or partial number
    const double t12 = ENERGY ? wave_sum(s12) : __builtin_nan(""), t6 = ENERGY ? wave_sum(s6) : __builtin_nan("");
    if (lane == 0) {
        double *w = a.wg_part + 2 * (((size_t)by * gridDim.x + bx) * W + wv);
        w[0] = t12;
        w[1] = t6;
    }
#ifdef LJMD_WAVE_TRACE
    if (lane == 0) trace_store((unsigned)((blockIdx.y * gridDim.x + blockIdx.x) * W + wv), tr0, tr1, tr2, tr_heavy, blockIdx.y * gridDim.x + blockIdx.x);
#endif
}

// ===========================================================================
// K2 far pass of the mixed-precision mode (LJMD_PRECISION_FP32_FORCE): the same Newton-3 rotation
// scheme, pair arithmetic in fp32.  Only tile pairs whose boxes are farther apart than r_split
// come here (mask_far); everything closer -- where the forces are large -- stays in the fp64 kernel.
//   * coordinates are tile-relative: offset = (float)(x - tile centre); the centre difference of a
//     (row tile, column tile) pair, minus the periodic image shift when it is uniform, is formed in
//     fp64, rounded once to fp32 and folded into the row offsets, so a pair costs 3 fp32 subtractions;
//   * 1/r^2 = v_rcp_f32 (1 ulp); per column tile the partial accelerations and the two energy sums
//     are accumulated in fp32 (<= 256 terms) and then added to fp64 accumulators;
//   * relative error of a far pair's force ~1e-7, absolute < 1e-11 (|f| < 24 r_split^-7).
// ===========================================================================
__device__ __forceinline__ float dpp_rotate_f32(float v)
{
    const int i = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x13C, 0xF, 0xF, false);
    return __int_as_float(i);
}

// GEN: bit k set = axis k needs the general minimum image; clear = its common image shift is already folded
// into the row offsets (0 = the former UNIFORM, 7 = all general)
// VFAR: every pair of the pass has r^2 > 2^(26/3) (r > 20.16 sigma), i.e. u^3 = r^-6 < 2^-26.  Then 2 u^6 < ulp(u^3) / 2 and
// fmaf(2, u^6, -u^3) IS -u^3, bit for bit: the loop leaves u^6 out -- three of the fourteen instructions behind the cutoff
// test -- and the accelerations keep their bits.  What changes is the energy sum S12 = sum u^6, which loses the terms
// beyond 20.16 sigma: n 4 pi rho / (9 r^9) = 2e-12 n, 3e-13 of epot (the mixed mode's own fp32 noise is 7e-11).
// (An INNER form -- no cutoff test where the boxes prove every pair inside -- was built twice and is 3.4 / 4.6 % SLOWER although
// it issues one VALU and three SALU instructions less per pair: profiles/r03_cluster_passes.txt, r04_f32_far_kernel_forms.txt.)
template <int GEN, bool ENERGY, bool VFAR = false>
__device__ __forceinline__ void pair_n3_f32(float xi, float yi, float zi, float xj, float yj, float zj,
                                            float Lf, float invLf, float rc2f,
                                            float &ax, float &ay, float &az, float &jx, float &jy, float &jz,
                                            float &s12, float &s6)
{
    float dx = xi - xj, dy = yi - yj, dz = zi - zj;
    if constexpr (GEN & 1) dx = fmaf(-Lf, __builtin_rintf(dx * invLf), dx);
    if constexpr (GEN & 2) dy = fmaf(-Lf, __builtin_rintf(dy * invLf), dy);
    if constexpr (GEN & 4) dz = fmaf(-Lf, __builtin_rintf(dz * invLf), dz);
    const float r2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    if (r2 < rc2f) {
        const float u = __builtin_amdgcn_rcpf(r2);
        const float u3 = u * u * u;
        float g;
        if constexpr (VFAR) {
            // (GEN == 0: the caller takes sum u^3 of the pass from its force sums -- below, in pair_n3_f32_kernel -- and nothing is added here)
            if constexpr (ENERGY && GEN != 0) s6 += u3;
            g = -u3 * u;                                  // = fmaf(2, u^6, -u^3) * u, exactly
        } else {
            const float u6 = u3 * u3;
            if constexpr (ENERGY) {
                s12 += u6;
                s6 += u3;
            }
            g = fmaf(2.0f, u6, -u3) * u;
        }
        ax = fmaf(g, dx, ax);
        ay = fmaf(g, dy, ay);
        az = fmaf(g, dz, az);
        jx = fmaf(-g, dx, jx);
        jy = fmaf(-g, dy, jy);
        jz = fmaf(-g, dz, jz);
    }
}

template <int GEN, bool MASKED, bool ENERGY, bool VFAR>
__device__ __forceinline__ void column_loop_f32(const float (&px)[kRowTiles], const float (&py)[kRowTiles],
                                                const float (&pz)[kRowTiles], float (&fx)[kRowTiles],
                                                float (&fy)[kRowTiles], float (&fz)[kRowTiles],
                                                float xj, float yj, float zj, unsigned mb,
                                                float Lf, float invLf, float rc2f,
                                                float &jx, float &jy, float &jz, float &s12, float &s6)
{
    // The per-pair `if (r2 < rc2f)` stays an exec-mask region and the loop stays rolled: measured against branch-free
    // forms (u = med3(1/r^2, 0, (rc^2 - r^2) 2^60) or a select: +13 ... +30 %), one wave-uniform branch per pair or per
    // step (+9 % / +40 %), the column offsets parked in LDS instead of rotating (+-0), 2 / 8 / 64 unrolled steps
    // (+2 % / +-0 / +15 %), 4 or 6 waves per SIMD (+1 % / +2 %), x and y packed into v_pk_add / v_pk_fma_f32 by hand (+4 %
    // for a single packed instruction): profiles/r04_f32_far_kernel_forms.txt.  The loop runs at
    // the sum of its instructions' issue costs (tools/ubench_f32mix.hip): 29 % of its wave-level pair evaluations have all
    // 64 lanes outside the cutoff and skip 13 of their 22 instructions through the region's s_cbranch_execz.
    for (int s = 0; s < kTile; ++s) {
#pragma unroll
        for (int k = 0; k < kRowTiles; ++k)
            if (!MASKED || ((mb >> k) & 1u))
                pair_n3_f32<GEN, ENERGY, VFAR>(px[k], py[k], pz[k], xj, yj, zj, Lf, invLf, rc2f, fx[k], fy[k], fz[k],
                                     jx, jy, jz, s12, s6);
        xj = dpp_rotate_f32(xj); yj = dpp_rotate_f32(yj); zj = dpp_rotate_f32(zj);
        jx = dpp_rotate_f32(jx); jy = dpp_rotate_f32(jy); jz = dpp_rotate_f32(jz);
    }
}

// 5 waves per SIMD (96 VGPRs; 20 spilled, none of them in the rotation loop): 11.73 -> 11.59 ms at n = 262144; 6: +-0, 8: +8 %
template <bool ENERGY>
__global__ __launch_bounds__(kTile, 5) void pair_n3_f32_kernel(N3Args a)
{
    const int lane = threadIdx.x;                              // one wave per workgroup (see pair_n3_kernel)
    unsigned bx = blockIdx.x, by = blockIdx.y;                 // XCD-aware work mapping, as in pair_n3_kernel
    if (a.xcd_remap) {
        const unsigned id = blockIdx.y * gridDim.x + blockIdx.x, per = gridDim.x >> 3, C = (unsigned)a.xcd_remap;
        const unsigned xcd = id & 7u, slot = id >> 3, s = slot % per;
        by = slot / per;
        bx = ((s / C) * 8u + xcd) * C + s % C;
    }
    const int Al = (int)bx;
    const bool active = Al < a.NGo;
    const int A = a.rank * a.NGo + Al;
    const size_t P = a.P;
    const double *own = a.pos + (size_t)a.rank * 3 * P;
    const float Lf = (float)a.L, invLf = (float)a.invL, rc2f = (float)a.rc2;

    // row tiles: centre (from the exact boxes), fp32 offsets, fp64 accumulators
    double cx[kRowTiles], cy[kRowTiles], cz[kRowTiles];
    float ox[kRowTiles], oy[kRowTiles], oz[kRowTiles];
    double ax[kRowTiles], ay[kRowTiles], az[kRowTiles];
    double s12 = 0.0, s6 = 0.0;
#pragma unroll
    for (int k = 0; k < kRowTiles; ++k) {
        const int tl = active ? kRowTiles * Al + k : 0;
        const double *bb = a.bbox + (size_t)(a.rank * a.TB + tl) * kBoxStride;
        cx[k] = 0.5 * (bb[0] + bb[3]);
        cy[k] = 0.5 * (bb[1] + bb[4]);
        cz[k] = 0.5 * (bb[2] + bb[5]);
        const size_t slot = (size_t)tl * kTile + lane;
        ox[k] = (float)(own[slot] - cx[k]);
        oy[k] = (float)(own[P + slot] - cy[k]);
        oz[k] = (float)(own[2 * P + slot] - cz[k]);
        ax[k] = ay[k] = az[k] = 0.0;
    }

    const int d0 = (int)by * a.dchunk;
    const int d1 = active ? min(d0 + a.dchunk, a.Dmax + 1) : d0;
    for (int d = d0; d < d1; ++d) {
        int B = A + d;
        if (B >= a.NG) B -= a.NG;
        const bool own_pair = (d == 0) || (2 * d < a.NG) || (2 * d == a.NG && A < B);
        for (int l = 0; l < kRowTiles; ++l) {
            const int c = kRowTiles * B + l;
            const size_t blk = (size_t)c * a.CS + (size_t)(a.by_group ? Al : d);   // N3Args::slab_j
            unsigned desc = 0;
            if (own_pair && d != 0)                        // the diagonal group is always a NEAR (fp64) pair
                desc = (unsigned)__builtin_amdgcn_readfirstlane((int)a.desc[(size_t)Al * a.T + c]);   // built from mask_far
            const unsigned mb = desc & 15u;
            if (mb == 0) {
                if (lane == 0) a.flag_j[blk] = 0;
                continue;
            }
            const double *cbx = a.bbox + (size_t)c * kBoxStride;
            const double ccx = 0.5 * (cbx[0] + cbx[3]), ccy = 0.5 * (cbx[1] + cbx[4]), ccz = 0.5 * (cbx[2] + cbx[5]);
            // image class from the pass descriptor (tile_class_kernel): per axis common image (its shift folded into the
            // row offsets below) or general; two or more general axes -> all general (a folded shift on an axis that is
            // then also treated generally is harmless)
            const int nu = (int)((desc >> 4) & 31u);
            const int gen = nu == 27 ? 1 : nu == 28 ? 2 : nu == 29 ? 4 : (nu == 7 || nu == 30) ? 7 : 0;
            const double sx = (double)((int)((desc >> 11) & 7u) - 2) * a.L;
            const double sy = (double)((int)((desc >> 14) & 7u) - 2) * a.L;
            const double sz = (double)((int)((desc >> 17) & 7u) - 2) * a.L;
            const int gj = (a.G == 1) ? 0 : c / a.TB;
            const double *cb = a.pos + (size_t)gj * 3 * P + (size_t)(c - gj * a.TB) * kTile + lane;
            const float xj = (float)(cb[0] - ccx), yj = (float)(cb[P] - ccy), zj = (float)(cb[2 * P] - ccz);

            // row offsets in the column tile's frame: (x_i - c_k) + ((c_k - c_c) - shift)
            float px[kRowTiles], py[kRowTiles], pz[kRowTiles], fx[kRowTiles], fy[kRowTiles], fz[kRowTiles];
#pragma unroll
            for (int k = 0; k < kRowTiles; ++k) {
                px[k] = ox[k] + (float)((cx[k] - ccx) - sx);
                py[k] = oy[k] + (float)((cy[k] - ccy) - sy);
                pz[k] = oz[k] + (float)((cz[k] - ccz) - sz);
                fx[k] = fy[k] = fz[k] = 0.0f;
            }
            float jx = 0.0f, jy = 0.0f, jz = 0.0f, t12 = 0.0f, t6 = 0.0f;
#define LJMD_LOOP32(GEN_, MASKED_, VFAR_)                                                                       \
    column_loop_f32<GEN_, MASKED_, ENERGY, VFAR_>(px, py, pz, fx, fy, fz, xj, yj, zj, mb, Lf, invLf, rc2f, jx, jy, jz, t12, t6)
#define LJMD_LOOP32_GEN(GEN_)                                                                                   \
    do {                                                                                                        \
        if (vfar) { if (all4) LJMD_LOOP32(GEN_, false, true); else LJMD_LOOP32(GEN_, true, true); }             \
        else      { if (all4) LJMD_LOOP32(GEN_, false, false); else LJMD_LOOP32(GEN_, true, false); }           \
    } while (0)
            const bool all4 = mb == kAllRows;
            const bool vfar = ((desc >> 28) & 1u) != 0;       // every pair beyond 20.16 sigma (tile_class)
            if (gen == 0)      LJMD_LOOP32_GEN(0);
            else if (gen == 1) LJMD_LOOP32_GEN(1);
            else if (gen == 2) LJMD_LOOP32_GEN(2);
            else if (gen == 4) LJMD_LOOP32_GEN(4);
            else               LJMD_LOOP32_GEN(7);
#undef LJMD_LOOP32_GEN
            if (ENERGY && vfar && gen == 0) {
                // sum u^3 of a VERY FAR pass without a general axis, from its force sums: there g = -u^4 and the displacement
                // is the plain difference d = p_i - q_j of the offsets the loop ran on, so
                //   sum u^3 = -sum g d.d = -( sum_i p_i . f_i + sum_j q_j . j_j ),   f_i = sum_j g d,  j_j = -sum_i g d
                // -- 15 FMAs per pass in place of one addition per pair (256 per lane).  q (offsets from the column tile's
                // own centre) is small and p ~ d: no cancellation; fp32 like the sum it replaces.  (After 64 rotations the
                // column particle and its sums are back in their home lane.)
                float w = fmaf(zj, jz, fmaf(yj, jy, xj * jx));
#pragma unroll
                for (int k = 0; k < kRowTiles; ++k) w = fmaf(pz[k], fz[k], fmaf(py[k], fy[k], fmaf(px[k], fx[k], w)));
                t6 = -w;
            }
#undef LJMD_LOOP32
#pragma unroll
            for (int k = 0; k < kRowTiles; ++k) {
                ax[k] += (double)fx[k];
                ay[k] += (double)fy[k];
                az[k] += (double)fz[k];
            }
            s12 += (double)t12;
            s6 += (double)t6;
            // the column-side sums of a far pass ARE fp32: their blocks are stored as such (half the slab traffic of this
            // kernel and of the reduction, which widens them exactly before it adds)
            float *o = reinterpret_cast<float *>(a.slab_j) + blk * (3 * kTile) + lane;
            o[0] = jx;
            o[kTile] = jy;
            o[2 * kTile] = jz;
            if (lane == 0) a.flag_j[blk] = 1;
        }
    }

    if (active) {
        double *si = a.slab_i + (size_t)by * 3 * P;
#pragma unroll
        for (int k = 0; k < kRowTiles; ++k) {
            const size_t slot = (size_t)(kRowTiles * Al + k) * kTile + lane;
            si[slot] = ax[k];
            si[P + slot] = ay[k];
            si[2 * P + slot] = az[k];
        }
    }
    const double r12 = ENERGY ? wave_sum(s12) : __builtin_nan(""), r6 = ENERGY ? wave_sum(s6) : __builtin_nan("");
    if (lane == 0) {
        double *w = a.wg_part + 2 * ((size_t)by * gridDim.x + bx);
        w[0] = r12;
        w[1] = r6;
    }
}

// ---------------------------------------------------------------------------
// Geometry pre-pass 1: exact axis-aligned bounding box of every 64-slot tile of the
// exchange buffer (NaN padding ignored).  One wave per tile; reads 24 N bytes.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_first(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// the periodic image of x nearest to the anchor; x itself (bit for bit) when |x - anchor| < L / 2
__device__ __forceinline__ double tile_frame(double x, double anchor, double L, double invL)
{
    return fma(-L, __builtin_rint((x - anchor) * invL), x);
}

__global__ __launch_bounds__(kBlock) void tile_boxes_kernel(GeometryArgs a)
{
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (t >= a.T) return;
    const int g = t / a.TB, tl = t - g * a.TB;
    const size_t o0 = (size_t)g * 3 * a.P + (size_t)tl * kTile + lane;
    const double *b = a.pos + o0;
    double x = b[0], y = b[a.P], z = b[2 * (size_t)a.P];
    if (a.pos_tc) {
        // Tile frame (Newton-3 path): a particle that has just crossed a face of the box sits at the other end of the
        // wrapped interval, and a plain bounding box of its tile would span the whole box on that axis -- the tile would
        // meet every column on that axis, lose its common-image and INNER passes, and (mixed mode) its fp32 offsets would
        // be ~L/2.  The pair kernels therefore read a copy in which every particle of a tile is in the periodic image
        // nearest to the first particle of its row group (x' = x - L rint((x - x0) / L): x itself unless the group straddles a
        // face); the box bounds those coordinates.  Forces depend on coordinate differences modulo L only.
        // (the anchor is the first particle of the tile's ROW GROUP: the RT tiles of a group share one frame, so that
        //  the group's box -- the union that decides the image class and INNER of its passes -- stays compact too)
        const double *g0 = a.pos + (size_t)g * 3 * a.P + (size_t)(tl / a.RT) * a.RT * kTile;
        x = tile_frame(x, g0[0], a.L, a.invL);
        y = tile_frame(y, g0[a.P], a.L, a.invL);
        z = tile_frame(z, g0[2 * (size_t)a.P], a.L, a.invL);
        double *c = a.pos_tc + o0;
        c[0] = x; c[a.P] = y; c[2 * (size_t)a.P] = z;
    }
    const double lx = wave_min(x), ly = wave_min(y), lz = wave_min(z);
    const double hx = wave_max(x), hy = wave_max(y), hz = wave_max(z);
    if (lane == 0) {
        double *o = a.bbox + (size_t)t * kBoxStride;
        o[0] = lx; o[1] = ly; o[2] = lz;
        o[3] = hx; o[4] = hy; o[5] = hz;
    }
}

// ---------------------------------------------------------------------------
// Geometry pre-pass 2: tile-pair mask.  Bit J of row I is cleared only when the boxes
// prove that every pair (i in I, j in J) has r^2 > rc^2 (1 + 1e-10) under the minimum
// image -- such pairs fail the reference's `rij2 < rc_square` test, so skipping them
// changes nothing.  One wave per (row tile, 64-column word); lane = column tile.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void tile_mask_kernel(GeometryArgs a)
{
    const int lane = threadIdx.x & 63;
    const int Il = blockIdx.y * kWavesPerBlock + (threadIdx.x >> 6);
    const int w = blockIdx.x;
    if (Il >= a.TB) return;
    const int I = a.rank * a.TB + Il;
    const int J = w * 64 + lane;
    bool keep = false, far = false;
    if (J < a.T) {
        const double d2 = tile_gap2(a.bbox, I, J, a.L);
        keep = !(d2 > a.rc2_skin) || (J == I);
        if (a.mask_far) {
            // mixed precision: fp64 for boxes closer than r_split and inside the own row group, fp32 beyond
            const bool near = (d2 <= a.rsplit2) || (J / a.RT == I / a.RT);
            far = keep && !near;
            keep = keep && near;
        }
    }
    if (a.mask_far) {
        const uint64_t far_word = __ballot(far);
        if (lane == 0) a.mask_far[(size_t)Il * a.W + w] = far_word;
    }
    const uint64_t word = __ballot(keep);
    if (lane == 0) a.mask[(size_t)Il * a.W + w] = word;
}

// ---------------------------------------------------------------------------
// Geometry pre-pass 3 (Newton-3 kernel): one 32-bit descriptor per (owned row group, column tile) -- everything the
// pair kernel needs to know about a pass, computed ONCE by one lane here instead of redundantly by the 64 lanes of the
// wave in the pair kernel's prologue (~120 fp64 VALU instructions per pass there, 2 % of the kernel):
//   bits  0..3   row tiles of the group whose mask bit for this column tile is set (tile_mask_kernel)
//   bits  4..8   loop variant nu (pair_n3): image class of the (row group, column tile) pair
//   bit   9      INNER: every pair provably inside the cutoff
//   bit   10     FULL: no padding slot in the row group or the column tile
//   bits 11..19  common image per axis, n + 2 in 3 bits each (shift = n L)
//   bit   20     CLUSTER: the pass runs cluster by cluster (n3_cluster_pass) with the direction / thresholds in desc2
//   bit   21     PERTILE: the row tiles are shifted by whole box lengths on axis (bits 22-23) for this pass, tile k by
//                (bits 24 + 2k .. 25 + 2k) - 1; the column tile's common image on that axis is the first active tile's
//   bits 22..27  (PERTILE clear) straddle passes, nu 27..30: bit 22 + q = axis q straddles a half-box distance, bit 25 + q =
//                that distance is (n + 1/2) L rather than (n - 1/2) L, n = the axis' image field (pair_disp)
// Same expressions as the former in-kernel classification; the row group's box is the union of its tiles' exact
// boxes (= min / max over its 256 particles).
// ---------------------------------------------------------------------------
// -> false: the pair kernel does not visit (row group Al, column tile c).  desc_far / desc2: NULL = not wanted.
// (RT = tiles per row group, a template parameter: the loops over the row tiles unroll and their small arrays stay in registers)
template <int RT>
__device__ __forceinline__ bool tile_class(const GeometryArgs &a, double invL, double rc2, int S, int Al, int c,
                                           unsigned &desc_out, unsigned *desc_far, float *desc2)
{
    {   // only the (row group, column group) pairs the pair kernel visits: offset d = (B - A) mod NG in 0 .. NG / 2,
        // the pair at exactly NG / 2 from its lower-numbered side (pair_n3_kernel's own_pair)
        const int NG = a.T / RT, A = a.rank * (a.TB / RT) + Al, B = c / RT;
        int d = B - A;
        if (d < 0) d += NG;
        if (!(d == 0 || 2 * d < NG || (2 * d == NG && (A < B || a.both_ties)))) return false;
    }
    double glo[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()};
    double ghi[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
    const double *cbx = a.bbox + (size_t)c * kBoxStride;
    unsigned mb = 0, mb_far = 0;
    bool vfar = true;
    for (int k = 0; k < RT; ++k) {
        const int tl = RT * Al + k, I = a.rank * a.TB + tl;
        const double *bb = a.bbox + (size_t)I * kBoxStride;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            glo[q] = fmin(glo[q], bb[q]);                      // fmin / fmax ignore the NaN of an all-padding tile
            ghi[q] = fmax(ghi[q], bb[3 + q]);
        }
        // the tile-pair test of tile_mask_kernel (same expressions): the bit is cleared only when the boxes prove every
        // pair to be outside the cutoff; mixed precision splits the kept pairs into NEAR (fp64) and FAR (fp32)
        double d2 = 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const double g = axis_gap(bb[q] - cbx[3 + q], bb[3 + q] - cbx[q], a.L);
            d2 += g * g;
        }
        bool keep = !(d2 > a.rc2_skin) || (c == I);
        if (desc_far) {
            const bool near = (d2 <= a.rsplit2) || (c / RT == I / RT);
            mb_far |= (unsigned)(keep && !near) << k;
            if (keep && !near && !(d2 > a.rvfar2)) vfar = false;   // a far row tile closer than the VERY FAR radius
            keep = keep && near;
        }
        mb |= (unsigned)keep << k;
    }
    if (mb == 0 && mb_far == 0) {                            // nothing inside the cutoff: the pair kernels look no further
        desc_out = 0;
        if (desc_far) desc_far[(size_t)Al * a.T + c] = 0;
        return true;
    }
    double sx = 0.0, sy = 0.0, sz = 0.0;
    const double lo[3] = {glo[0] - cbx[3], glo[1] - cbx[4], glo[2] - cbx[5]};
    const double hi[3] = {ghi[0] - cbx[0], ghi[1] - cbx[1], ghi[2] - cbx[2]};
    const bool ux = uniform_image(lo[0], hi[0], a.L, invL, sx);
    const bool uy = uniform_image(lo[1], hi[1], a.L, invL, sy);
    const bool uz = uniform_image(lo[2], hi[2], a.L, invL, sz);
    int nu = (ux ? 0 : 1) | (uy ? 0 : 2) | (uz ? 0 : 4);
    // ONE axis without a common image for the row GROUP (its box, 4 tiles, straddles +-L/2 against the column tile): the
    // row TILES, half as wide, mostly have one each.  Then the column tile takes the image of the first active row tile
    // and the other row tiles are shifted by -1 / 0 / +1 box lengths for this pass (pair_n3_kernel: shift_rows): the pass
    // runs the plain loop (3 instructions per pair less than the general minimum image) and may run cluster by cluster.
    // 19-23 % of the passes of the liquid have a general axis; about half of them go this way.
    int pq = -1;                                              // the axis, -1 = none
    int pm[4] = {0, 0, 0, 0};                                  // per row tile: image relative to the first active tile's
    if (a.pertile_images && !desc_far && RT > 1 && mb != 0 && (nu == 1 || nu == 2 || nu == 4)) {
        const int q = nu == 1 ? 0 : nu == 2 ? 1 : 2;
        int nk[4] = {0, 0, 0, 0}, nbase = 0;
        bool ok = true, have = false;
        for (int k = 0; k < RT && ok; ++k) {
            if (!((mb >> k) & 1u)) continue;
            const double *bb = a.bbox + (size_t)(a.rank * a.TB + RT * Al + k) * kBoxStride;
            double sk = 0.0;
            ok = uniform_image(bb[q] - cbx[3 + q], bb[3 + q] - cbx[q], a.L, invL, sk);
            nk[k] = (int)__builtin_rint(sk * invL);
            if (ok && !have) { nbase = nk[k]; have = true; }
            ok = ok && nk[k] - nbase >= -1 && nk[k] - nbase <= 1;
        }
        if (ok && have) {
            pq = q;
            for (int k = 0; k < RT; ++k) pm[k] = ((mb >> k) & 1u) ? nk[k] - nbase : 0;
            const double sb = (double)nbase * a.L;
            if (q == 0) sx = sb; else if (q == 1) sy = sb; else sz = sb;
            nu = 0;                                             // a common image on every axis now
        }
    }
    // box of row tile k on axis q as the pass sees it
    auto row_lo = [&](const double *bb, int k, int q) { return bb[q] - (q == pq ? (double)pm[k] * a.L : 0.0); };
    auto row_hi = [&](const double *bb, int k, int q) { return bb[3 + q] - (q == pq ? (double)pm[k] * a.L : 0.0); };
    const bool group_full = (RT * Al + RT) * kTile <= S;
    const bool full = group_full && ((c - (a.G == 1 ? 0 : c / a.TB) * a.TB) + 1) * kTile <= S;
    // INNER: every pair of every ACTIVE row tile provably inside the cutoff (farthest corners of the exact tile boxes;
    // the group's box, the union, proves less: 33 % instead of 44 % of the passes of the bench configuration)
    bool inner = nu == 0 && full && mb != 0;
    for (int k = 0; k < RT && inner; ++k) {
        if (!((mb >> k) & 1u)) continue;
        const double *bb = a.bbox + (size_t)(a.rank * a.TB + RT * Al + k) * kBoxStride;
        const double sh[3] = {sx, sy, sz};
        double far2 = 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const double f = fmax(fabs((row_lo(bb, k, q) - cbx[3 + q]) - sh[q]), fabs((row_hi(bb, k, q) - cbx[q]) - sh[q]));
            far2 += f * f;
        }
        inner = far2 < rc2 * (1.0 - 1e-10);
    }
    // Axes without a common image: the straddle form (pair_disp) whenever the range of differences lies strictly inside
    // ((h - 1) L, (h + 1) L) around the half-integer h = n +- 1/2 it contains -- always, with tile frames, unless a tile is
    // wider than half the box -- and every such axis of the pass qualifies; else the general minimum image on all axes.
    unsigned straddle_axes = 0, straddle_signs = 0;
    if (nu != 0) {
        bool ok = true;
        for (int q = 0; q < 3 && ok; ++q) {
            if (!((nu >> q) & 1)) continue;
            const double tlo = lo[q] * invL, thi = hi[q] * invL;
            const double n = __builtin_rint(0.5 * (tlo + thi));         // = sh[q] / L (uniform_image)
            const bool up = 0.5 * (tlo + thi) >= n;                       // the half-integer on this side of n
            const double h = up ? n + 0.5 : n - 0.5;
            ok = fabs(n) <= 2.0 && tlo > h - 1.0 + 1e-9 && thi < h + 1.0 - 1e-9;
            straddle_axes |= 1u << q;
            straddle_signs |= (up ? 1u : 0u) << q;
        }
        if (ok) {
            nu = (straddle_axes == 1u) ? 27 : (straddle_axes == 2u) ? 28 : (straddle_axes == 4u) ? 29 : 30;
        } else {
            straddle_axes = straddle_signs = 0;
            sx = sy = sz = 0.0;                                           // rndne finds every image itself
            nu = 7;
        }
    }
    if (nu == 0) {
        const int nz = (sx != 0.0 ? 1 : 0) | (sy != 0.0 ? 2 : 0) | (sz != 0.0 ? 4 : 0);
        nu = nz == 0 ? 8 : nz == 1 ? 16 : nz == 2 ? 17 : nz == 4 ? 18 : 0;   // none / one axis / several
    }
    // shift = n L with |n| <= 2 (uniform_image): recover n exactly
    const int nx = (int)__builtin_rint(sx * invL), ny = (int)__builtin_rint(sy * invL), nzs = (int)__builtin_rint(sz * invL);
    unsigned cls = ((unsigned)nu << 4) | ((unsigned)inner << 9) | ((unsigned)full << 10) |
                   ((unsigned)(nx + 2) << 11) | ((unsigned)(ny + 2) << 14) | ((unsigned)(nzs + 2) << 17);
    if (pq >= 0) {                                            // bit 21: per-tile images; 22-23: the axis; 24..31: image + 1 per row tile
        cls |= (1u << 21) | ((unsigned)pq << 22);
        for (int k = 0; k < 4; ++k) cls |= (unsigned)(pm[k] + 1) << (24 + 2 * k);
    } else {
        cls |= (straddle_axes << 22) | (straddle_signs << 25);   // (bit 21 clear: these bits name the straddling axes)
    }
    // Cluster pass (pair_n3_kernel: n3_cluster_pass): a pass at the cutoff boundary with a common image on every axis, no
    // padding slot, not inside the diagonal group.  desc2 = the unit direction n from the row group to the (shifted)
    // column tile, and per row tile k the threshold  thr_k = max over the tile's box of n.x  +  rc  +  margin:
    // a column particle with  n.(xj + s) > thr_k  is farther than rc from every particle of row tile k.  The margin
    // (1e-3) covers the fp32 roundings of n, of the kernel's projection and of thr itself (each < 1e-5 at L = 110).
    if (desc2 && mb != 0 && !inner && full && c / RT != a.rank * (a.TB / RT) + Al &&
        (nu == 8 || nu == 16 || nu == 17 || nu == 18 || nu == 0)) {
        double dir[3], len2 = 0.0;
        const double sh[3] = {sx, sy, sz};
        if (pq >= 0) {                                         // the group's box as the pass sees it: its tiles shifted
            glo[pq] = __builtin_inf();
            ghi[pq] = -__builtin_inf();
            for (int k = 0; k < RT; ++k) {
                const double *bb = a.bbox + (size_t)(a.rank * a.TB + RT * Al + k) * kBoxStride;
                glo[pq] = fmin(glo[pq], row_lo(bb, k, pq));
                ghi[pq] = fmax(ghi[pq], row_hi(bb, k, pq));
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            dir[q] = (0.5 * (cbx[q] + cbx[3 + q]) + sh[q]) - 0.5 * (glo[q] + ghi[q]);
            len2 += dir[q] * dir[q];
        }
        if (len2 > 1.0) {                                      // (boxes on top of each other: no boundary to speak of)
            const double inv = 1.0 / sqrt(len2);
            float nf[3];
            double nn = 0.0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                nf[q] = (float)(dir[q] * inv);
                nn += (double)nf[q] * (double)nf[q];
            }
            if (nn <= 1.0 + 1e-6) {
                float *o = desc2 + ((size_t)Al * a.T + c) * 8;
                o[0] = nf[0]; o[1] = nf[1]; o[2] = nf[2];
                const double rc = sqrt(rc2);
                for (int k = 0; k < RT; ++k) {
                    const double *bb = a.bbox + (size_t)(a.rank * a.TB + RT * Al + k) * kBoxStride;
                    double supp = 0.0;
#pragma unroll
                    for (int q = 0; q < 3; ++q) supp += fmax((double)nf[q] * row_lo(bb, k, q), (double)nf[q] * row_hi(bb, k, q));
                    o[3 + k] = (float)(supp + rc + 1e-3);
                }
                for (int k = RT; k < 4; ++k) o[3 + k] = 0.0f;
                o[7] = 0.0f;
                // ... and only where some cluster CAN be skipped: the largest projection the column tile's box allows
                // must exceed the threshold of at least one active row tile
                double colmax = 0.0;
#pragma unroll
                for (int q = 0; q < 3; ++q) colmax += fmax((double)nf[q] * (cbx[q] + sh[q]), (double)nf[q] * (cbx[3 + q] + sh[q]));
                bool useful = false;
                for (int k = 0; k < RT; ++k) useful = useful || (((mb >> k) & 1u) && colmax > (double)o[3 + k]);
                if (useful) cls |= 1u << 20;
            }
        }
    }
    desc_out = mb | cls;
    // far pass: bit 28 (free here: no per-tile images in the mixed mode) = VERY FAR
    // (FULL passes only: the kernel multiplies offsets with force sums, and a padding slot's offset is NaN)
    if (desc_far) desc_far[(size_t)Al * a.T + c] = mb_far | cls | ((unsigned)(vfar && full && mb_far != 0) << 28);
    return true;
}

template <int RT>
__global__ __launch_bounds__(kBlock) void tile_class_kernel(GeometryArgs a, double invL, double rc2, int S, unsigned *desc,
                                                            unsigned *desc_far, float *desc2)
{
    const int c = blockIdx.x * kBlock + threadIdx.x;           // column tile (global)
    const int Al = blockIdx.y;                                 // owned row group
    if (c >= a.T) return;
    unsigned d = 0;
    if (tile_class<RT>(a, invL, rc2, S, Al, c, d, desc_far, desc2)) desc[(size_t)Al * a.T + c] = d;
}

// ---------------------------------------------------------------------------
// K1: drift + wrap + first half-kick + unwrapped update, one thread per slot.
//   r  = (r + v*dt) + a*dt_square_half        verlet.f90:58-60   (left-to-right, unfused)
//   r  = r - L*floor(r*invL)                  geometry_pbc.f90:54-56
//   v  = v + a*dt_half                        verlet.f90:72-74
//   ru = ru + mic(r_new - r_old)              md_simulation_program.f90:341-351 (dnint)
// HBM-bound: reads r,v,a,ru (96 B) + writes r,v,ru (72 B) = 168 B per particle.
// ---------------------------------------------------------------------------
// PHASE 0: everything.  Multi-rank runs split it so that the position all-gather can start as early as
// possible: PHASE 1 = positions (r, wrap, ru), PHASE 2 = the velocity half-kick, which then runs
// while the all-gather is in flight on the communication stream.  Same arithmetic either way.
template <int PHASE, bool BOXES = false>
__global__ __launch_bounds__(kBlock) void drift_kick_kernel(IntegrateArgs a)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.rows) return;                     // rows = P is a multiple of the block size: whole waves only
    double rn[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const size_t o = (size_t)ax * a.P + i;
        const double v0 = a.v[o], acc = a.a[o];
        if constexpr (PHASE != 2) {
            const double r0 = a.r[o];
            double r1 = (r0 + v0 * a.dt) + acc * a.dt_sq_half;
            r1 = r1 - a.L * __builtin_floor(r1 * a.invL);
            double d = r1 - r0;
            d = d - a.L * __builtin_round(d * a.invL);
            a.r[o] = r1;
            a.ru[o] = a.ru[o] + d;
            rn[ax] = r1;
        }
        if constexpr (PHASE != 1) a.v[o] = v0 + acc * a.dt_half;
    }
    if constexpr (BOXES && PHASE != 2) {
        // single rank: the wave holds exactly one tile -- emit its bounding box here (same values, same
        // reductions as tile_boxes_kernel) and save that launch
        if (a.pos_tc) {                                       // Newton-3 path: tile frame, as in tile_boxes_kernel
            // the block's four waves are four consecutive tiles = whole row groups (RT = 1, 2 or 4): the anchor of a
            // group, its first tile's first particle, comes from the neighbouring wave through LDS
            __shared__ double anchor[kWavesPerBlock][3];
            const int w = threadIdx.x >> 6;
            if ((threadIdx.x & 63) == 0) {
                anchor[w][0] = rn[0]; anchor[w][1] = rn[1]; anchor[w][2] = rn[2];
            }
            __syncthreads();
            const int w0 = (w / a.RT) * a.RT;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                rn[ax] = tile_frame(rn[ax], anchor[w0][ax], a.L, a.invL);
                a.pos_tc[(size_t)ax * a.P + i] = rn[ax];
            }
        }
        const double lx = wave_min(rn[0]), ly = wave_min(rn[1]), lz = wave_min(rn[2]);
        const double hx = wave_max(rn[0]), hy = wave_max(rn[1]), hz = wave_max(rn[2]);
        if ((threadIdx.x & 63) == 0) {
            double *o = a.bbox + (size_t)(i >> 6) * kBoxStride;
            o[0] = lx; o[1] = ly; o[2] = lz;
            o[3] = hx; o[4] = hy; o[5] = hz;
        }
    }
}


// ---------------------------------------------------------------------------
// K3a: deterministic reduction of the partial-acceleration slabs (fixed order: row-side
// slices, then column-side blocks by ascending owned row group).  See ReduceArgs.
// grid = (P / 256, number of fpart blocks).
// ---------------------------------------------------------------------------
template <bool N3>
__global__ __launch_bounds__(kBlock) void reduce_forces_kernel(ReduceArgs a)
{
    // 64 slots (one tile) per workgroup; the 4 waves split the terms of every slot's sum 4 ways
    // (wave q takes slices / row groups q, q+4, ...) and the four partial sums are combined through LDS
    // in fixed order: 4x the loads in flight of a one-thread-per-slot loop, still bitwise reproducible.
    __shared__ double part[kWavesPerBlock][3][kTile];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * kTile + lane;                    // slot inside the block
    const int g = (gridDim.y == 1) ? a.rank : (int)blockIdx.y;  // whose particles
    double s[3] = {0.0, 0.0, 0.0};
    if (g == a.rank) {
        for (int c = q; c < a.nslab; c += kWavesPerBlock) {
            const double *sl = a.slab + (size_t)c * 3 * a.P + i;
            s[0] += sl[0];
            s[1] += sl[a.P];
            s[2] += sl[2 * (size_t)a.P];
        }
    }
    if constexpr (N3) {
        // The column tile's blocks lie together (N3Args::slab_j): block j of [0, CS) is added when its flag is set, wave q
        // taking j = q, q + 4, ... in ascending order.  The flags of 64 blocks are read at once (one byte per lane, a
        // ballot, the next 64 requested before the current ones are used), and the blocks the ballot names are loaded four
        // at a time: a stream of independent loads, nothing is read for a block nobody wrote.
        const int c = g * a.TB + blockIdx.x;                    // global column tile (= this workgroup's tile)
        const int qs = __builtin_amdgcn_readfirstlane(q);
        auto add_blocks = [&](const auto *slab, const unsigned char *flags, int CS) {
            constexpr int U = 4;
            const size_t base = (size_t)c * CS;
            unsigned f = lane < CS ? flags[base + lane] : 0u;
            for (int j0 = 0; j0 < CS; j0 += kTile) {
                unsigned long long m = __ballot(f != 0u) & (0x1111111111111111ull << qs);
                const int jn = j0 + kTile + lane;               // (requested after f is in: the counter of loads is in order)
                const unsigned fn = jn < CS ? flags[base + jn] : 0u;
                while (m) {
                    double v[U][3];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        v[u][0] = v[u][1] = v[u][2] = 0.0;      // s + 0.0 == s: fewer than U blocks left
                        if (m) {
                            const int j = j0 + __builtin_ctzll(m);
                            m &= m - 1;
                            const auto *b = slab + (base + j) * (3 * kTile) + lane;
                            v[u][0] = (double)b[0];
                            v[u][1] = (double)b[kTile];
                            v[u][2] = (double)b[2 * kTile];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        s[0] += v[u][0];
                        s[1] += v[u][1];
                        s[2] += v[u][2];
                    }
                }
                f = fn;
            }
        };
        add_blocks(a.slab_j, a.flag_j, a.CS);
        if (a.slab_j2) add_blocks(a.slab_j2, a.flag_j2, a.CS2);   // fp32 far pass of the mixed-precision mode
    }
    part[q][0][lane] = s[0];
    part[q][1][lane] = s[1];
    part[q][2][lane] = s[2];
    __syncthreads();
    if (q == 0) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            double t = part[0][ax][lane];
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) t += part[w][ax][lane];
            a.fpart[(size_t)(gridDim.y == 1 ? 0 : g) * 3 * a.P + (size_t)ax * a.P + i] = t;
        }
    }
}

// ---------------------------------------------------------------------------
// K3b: restore the x24 prefactor (lj_potential_energy.f90:189-191), optionally apply the
// second half-kick (verlet.f90:86-88) and emit per-block partial sums of vx^2, vy^2, vz^2
// (verlet.f90:93-95 keeps the three sums separate).
// ---------------------------------------------------------------------------
template <bool KICK>
__global__ __launch_bounds__(kBlock) void kick_kernel(IntegrateArgs a)
{
    __shared__ double red[3 * kWavesPerBlock];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    double k2[3] = {0.0, 0.0, 0.0};
    if (i < a.rows) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const size_t o = (size_t)ax * a.P + i;
            const double acc = 24.0 * a.fsum[o];
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
        const double vx = a.v[i], vy = a.v[a.P + i], vz = a.v[2 * (size_t)a.P + i];
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
// K4 pre-stage for large grids: block b folds the contiguous slice b of the per-workgroup pair sums
// into one pair (fixed order), so that the single-block finalize below reads at most kFoldBlocks pairs.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void fold_partials_kernel(const double *wg_part, double *folded, int n_wg)
{
    __shared__ double red[2 * kWavesPerBlock];
    const int per = (n_wg + gridDim.x - 1) / gridDim.x;
    const int w0 = blockIdx.x * per, w1 = min(w0 + per, n_wg);
    double v[2] = {0.0, 0.0};
    for (int w = w0 + threadIdx.x; w < w1; w += kBlock) {
        v[0] += wg_part[2 * (size_t)w];
        v[1] += wg_part[2 * (size_t)w + 1];
    }
    block_sum<2>(v, red);
    if (threadIdx.x == 0) {
        folded[2 * blockIdx.x] = v[0];
        folded[2 * blockIdx.x + 1] = v[1];
    }
}

// ---------------------------------------------------------------------------
// K4: one block folds the per-workgroup partials into ONE partial record of this
// rank for this step, appended to the scalar ring at *ring_pos:
//   rec = { S12, S6, Kx, Ky, Kz, 0, 0, 0 }    (kPartialStride doubles)
// The host (ljmd_combine_scalars) adds the ranks in rank order and applies the
// prefactors and tail constants.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void finalize_body(const FinalizeArgs &a, double *red /* [5 * kWavesPerBlock] */)
{
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    // (unrolled: the loads of eight terms are in flight together -- up to 80 terms per thread, and one after the other
    //  they cost a memory round trip each: 20 us at 16 384 partials; the additions keep their order)
#pragma unroll 8
    for (int w = threadIdx.x; w < a.n_wg; w += kBlock) {
        v[0] += a.wg_part[2 * (size_t)w];
        v[1] += a.wg_part[2 * (size_t)w + 1];
    }
#pragma unroll 4
    for (int b = threadIdx.x; b < a.n_ke; b += kBlock) {
        if (a.ke_tile) {
            // per-TILE sums (tile_tail_kernel): the 256-slot block's partial as kick_kernel forms it -- its four waves'
            // sums added in order -- so that the record has the same bits whichever kernels produced it
            const double *w = a.ke_tile + 3 * (size_t)(kWavesPerBlock * b);
            v[2] += ((w[0] + w[3]) + w[6]) + w[9];
            v[3] += ((w[1] + w[4]) + w[7]) + w[10];
            v[4] += ((w[2] + w[5]) + w[8]) + w[11];
        } else {
            v[2] += a.ke_part[3 * (size_t)b];
            v[3] += a.ke_part[3 * (size_t)b + 1];
            v[4] += a.ke_part[3 * (size_t)b + 2];
        }
    }
    block_sum<5>(v, red);
    if (threadIdx.x == 0) {
        const unsigned pos = *a.ring_pos;
        double *rec = a.ring + (size_t)(pos % a.ring_cap) * kPartialStride;
        rec[0] = v[0] * a.pair_scale;   // exact: the scale is 0.5 or 1
        rec[1] = v[1] * a.pair_scale;
        rec[2] = v[2];
        rec[3] = v[3];
        rec[4] = v[4];
        rec[5] = 0.0;
        rec[6] = 0.0;
        rec[7] = 0.0;
        *a.ring_pos = pos + 1;
    }
}

__global__ __launch_bounds__(kBlock) void finalize_kernel(FinalizeArgs a)
{
    __shared__ double red[5 * kWavesPerBlock];
    finalize_body(a, red);
}

// K3b + K4 in one launch (small and medium systems, where every launch costs ~4 us of the step): the kick
// kernel's blocks take a ticket when their kinetic-energy partial is out; the block that draws the last
// ticket runs the finalize body -- the same code, hence the same summation order and the same bits.
template <bool KICK>
__global__ __launch_bounds__(kBlock) void kick_finalize_kernel(IntegrateArgs a, FinalizeArgs f)
{
    __shared__ double red[5 * kWavesPerBlock];
    __shared__ bool last;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    double k2[3] = {0.0, 0.0, 0.0};
    if (i < a.rows) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const size_t o = (size_t)ax * a.P + i;
            const double acc = 24.0 * a.fsum[o];
            a.a[o] = acc;
            if constexpr (KICK) {
                const double v1 = a.v[o] + acc * a.dt_half;
                a.v[o] = v1;
                k2[ax] = v1 * v1;
            }
        }
    }
    if constexpr (KICK) block_sum<3>(k2, red);
    if (threadIdx.x == 0) {
        if constexpr (KICK) {
            double *w = a.ke_part + 3 * (size_t)blockIdx.x;
            w[0] = k2[0];
            w[1] = k2[1];
            w[2] = k2[2];
        }
        __threadfence();                                   // this block's partial is visible device-wide ...
        last = atomicAdd(a.ticket, 1u) == gridDim.x - 1;   // ... before its ticket is
    }
    __syncthreads();
    if (!last) return;
    __threadfence();                                       // the other blocks' partials, fresh from L2
    finalize_body(f, red);
    if (threadIdx.x == 0) *a.ticket = 0u;                  // ready for the next launch (stream-ordered)
}

// ---------------------------------------------------------------------------
// Small single-rank systems (the reference's own range, BASELINE configs 1-2): a step is bound by the latency CHAIN of
// its dependent launches -- five kernels of 4-19 us at n = 4096 -- not by any of them.  tile_tail_kernel is everything
// behind the pair kernel AND, inside a batch of steps, the next step's K1, one 256-thread block per tile:
//   K3a  the tile's partial accelerations, the four waves splitting the terms as reduce_forces_kernel does;
//   K3b  x24, second half-kick, sum v^2 per tile (waves 0 .. 2: one axis each; lj_potential_energy.f90:189-191, verlet.f90:86-95);
//   K1'  DRIFT: the NEXT step's drift + wrap + first half-kick + unwrapped update of the tile's 64 particles
//        (verlet.f90:58-74, md_simulation_program.f90:341-351 -- a particle's K1 reads only its own r, v, a), the tile's
//        coherent copy and its bounding box, as drift_kick_kernel<0, true> writes them;
//   K4   ONE ticket per block; the block that draws the last one folds the step's record (finalize_body).
// With the pass descriptors worked out inside the pair kernel (N3Args::inline_class) a step is TWO launches.  Same
// expressions, same summation orders as the separate kernels: bit-identical results
// (tests/test_gpu_parity.py::test_fused_launches_are_bitwise_equal_to_the_separate_kernels).  A first attempt that put
// the pair loop into the same launch (tickets per tile, one device-scope fence pair per WAVE) was 3.4x slower than five
// launches: profiles/r03_small_n_single_launch_negative.txt.
// ---------------------------------------------------------------------------
// The step record (K4) is off the critical path when another step of the batch follows: with a.ticket == NULL this
// launch leaves its record to the NEXT launch, whose extra block (index = number of tiles) folds it (`prev`) while the
// tiles are being reduced -- no device-wide fence, no ticket, no last block.  The last launch of a batch draws tickets as
// before and its last block folds the pending record, then its own.  Records are appended in step order either way.
// TPB = tiles per block = tiles per Newton-3 row group (1, or 2: pair_n3_kernel<., 2, .> for small systems): the block's
// 4 TPB waves reduce TPB consecutive tiles side by side (wave 4 b + q = tile b's terms q, q + 4, ...: the split and the
// order of the one-tile form), then three waves of every tile integrate its three axes -- the tiles of a row group share
// the frame of their coherent copy (its first particle: tile_boxes_kernel), so the second tile takes the first one's anchor.
template <bool N3, bool KICK, bool DRIFT, int TPB>
__global__ __launch_bounds__(kBlock * TPB) void tile_tail_kernel(ReduceArgs ra, IntegrateArgs a, FinalizeArgs f, FinalizeArgs prev)
{
    __shared__ double part[kWavesPerBlock * TPB][3][kTile];
    __shared__ double red[5 * kWavesPerBlock];
    __shared__ bool last;
    if ((int)blockIdx.x == a.P / (kTile * TPB)) {               // the extra block: the previous step's record
        if (threadIdx.x >= kBlock) return;                      // (finalize_body is written for 256 threads)
        if (prev.ring) finalize_body(prev, red);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = wave & (kWavesPerBlock - 1), tb = wave / kWavesPerBlock;
    const int tile = blockIdx.x * TPB + tb;
    const int i = tile * kTile + lane;                          // slot
    // behind the reduction the waves q = 0, 1, 2 of a tile integrate its x, y, z (a particle's K3b / K1 arithmetic, the tile's
    // frame, box and sum v^2 are all per axis): their loads go out first and arrive while the partial sums are formed
    double v_in = 0.0, r_in = 0.0, ru_in = 0.0;
    if (q < 3) {
        const size_t o = (size_t)q * a.P + i;
        if constexpr (KICK || DRIFT) v_in = a.v[o];
        if constexpr (DRIFT) {
            r_in = a.r[o];
            ru_in = a.ru[o];
        }
    }
    double s[3] = {0.0, 0.0, 0.0};
    // The order of reduce_forces_kernel (single rank): the row-side slices, then the column-side blocks, wave q taking
    // the terms q, q + 4, ...  A block that was not written this step (flag 0) adds an exact zero here instead of being
    // skipped -- s + 0.0 == s -- so that the loads do not depend on the flags and are in flight together: the chain of
    // ~17 dependent load-and-add round trips per wave was half of this kernel's 17 us at n = 4096.
#pragma unroll 8
    for (int c = q; c < ra.nslab; c += kWavesPerBlock) {
        const double *sl = ra.slab + (size_t)c * 3 * ra.P + i;
        s[0] += sl[0];
        s[1] += sl[ra.P];
        s[2] += sl[2 * (size_t)ra.P];
    }
    if constexpr (N3) {
        // the tile's CS blocks, contiguous (N3Args::slab_j)
#pragma unroll 8
        for (int j = q; j < ra.CS; j += kWavesPerBlock) {
            const size_t blk = (size_t)tile * ra.CS + j;
            const bool on = ra.flag_j[blk] != 0;
            const double *b = ra.slab_j + blk * (3 * kTile) + lane;
            const double b0 = b[0], b1 = b[kTile], b2 = b[2 * kTile];      // (whatever an unwritten block holds is discarded)
            s[0] += on ? b0 : 0.0;
            s[1] += on ? b1 : 0.0;
            s[2] += on ? b2 : 0.0;
        }
    }
    part[wave][0][lane] = s[0];
    part[wave][1][lane] = s[1];
    part[wave][2][lane] = s[2];
    __syncthreads();
    __shared__ double anchor_s[3];                              // the row group's frame: its first tile's first particle
    double rn = 0.0;
    if (q < 3) {
        const int ax = q;
        double t = part[kWavesPerBlock * tb][ax][lane];
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) t += part[kWavesPerBlock * tb + w][ax][lane];
        const size_t o = (size_t)ax * a.P + i;
        const double acc = 24.0 * t;                            // kick_kernel
        a.a[o] = acc;
        double vel = v_in, k2 = 0.0;
        if constexpr (KICK) {
            vel = vel + acc * a.dt_half;
            k2 = vel * vel;
        }
        if constexpr (DRIFT) {                                  // drift_kick_kernel<0> of the next step
            const double r0 = r_in;
            double r1 = (r0 + vel * a.dt) + acc * a.dt_sq_half;
            r1 = r1 - a.L * __builtin_floor(r1 * a.invL);
            double d = r1 - r0;
            d = d - a.L * __builtin_round(d * a.invL);
            a.r[o] = r1;
            a.ru[o] = ru_in + d;
            rn = r1;
            vel = vel + acc * a.dt_half;
        }
        if constexpr (KICK || DRIFT) a.v[o] = vel;
        if constexpr (KICK) {
            const double ks = wave_sum(k2);
            if (lane == 0) const_cast<double *>(f.ke_tile)[3 * (size_t)tile + ax] = ks;
        }
    }
    if constexpr (DRIFT) {
        if (a.pos_tc) {                                         // tile frame: the block is the row group
            double anchor = wave_first(rn);
            if constexpr (TPB > 1) {                            // the group's later tiles take the first one's anchor
                if (q < 3 && tb == 0 && lane == 0) anchor_s[q] = anchor;
                __syncthreads();
                if (q < 3) anchor = anchor_s[q];
            }
            if (q < 3) {
                rn = tile_frame(rn, anchor, a.L, a.invL);
                a.pos_tc[(size_t)q * a.P + i] = rn;
            }
        }
        if (q < 3) {
            const double lo = wave_min(rn), hi = wave_max(rn);
            if (lane == 0) {
                double *o = a.bbox + (size_t)tile * kBoxStride;
                o[q] = lo;
                o[3 + q] = hi;
            }
        }
    }
    __syncthreads();                                       // the three waves' stores are out before thread 0 fences and draws the ticket
    if (threadIdx.x >= kBlock) return;                     // (the record is folded by 256 threads: finalize_body)
    if (!a.ticket) return;                                 // this step's record: the next launch's extra block
    if (threadIdx.x == 0) {
        __threadfence();                                   // this block's partial is visible device-wide ...
        last = atomicAdd(a.ticket, 1u) == gridDim.x - 1;   // ... before its ticket is
    }
    __syncthreads();
    if (!last) return;
    __threadfence();                                       // the other blocks' partials, fresh from L2
    if (prev.ring) {                                       // a record still pending from the step before: first
        finalize_body(prev, red);
        __syncthreads();
    }
    finalize_body(f, red);
    if (threadIdx.x == 0) *a.ticket = 0u;                  // ready for the next launch (stream-ordered)
}

// ===========================================================================
// Trajectory analysis pair pass: radial distribution histogram of ONE snapshot,
// the O(n^2) loop of the reference's compute_rdf (scripts/md_one_run_analysis.py:556-584).
// Bit-exact integer result: per pair the reference's numpy arithmetic with its roundings --
//   d = x_j - x_i ; d -= L * rint(d / L)      (np.rint = half-to-even; a true division)
//   r = sqrt(dx*dx + dy*dy + dz*dz)           (unfused, correctly rounded sqrt)
//   if r < rmax: hist[int(r / dr)] += 2       (i<j pairs, weight 2)
// evaluated here for every unordered pair (j > i) once, with weight 2 (the arithmetic is symmetric in i, j).
// One thread per i, j broadcast through scalar loads, histogram in LDS (ds_add_u32), one
// 64-bit global atomic per bin and block; integer sums are order independent.
// ===========================================================================
// rint(d / L) without the division: d * (1/L) is within 2 ulp of the true quotient, so its nearest integer is the
// reference's unless the product sits within 1e-9 of a half-integer -- then (practically never) the true
// division decides.  Same integer, hence the same bits downstream.
__device__ __forceinline__ double rdf_image(double d, double L, double invL)
{
    const double q = d * invL;
    double n = __builtin_rint(q);
    if (fabs(q - n) > 0.5 - 1e-9) n = __builtin_rint(d / L);
    return n;
}

__global__ __launch_bounds__(kBlock) void rdf_histogram_kernel(RdfArgs a)
{
    extern __shared__ unsigned lhist[];
    for (int b = threadIdx.x; b < a.nbins; b += kBlock) lhist[b] = 0u;
    __syncthreads();
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < a.n;
    const double xi = live ? a.x[i] : 0.0, yi = live ? a.y[i] : 0.0, zi = live ? a.z[i] : 0.0;
    // every unordered pair ONCE (j > i) with weight 2, as the reference counts (d_ji = -d_ij and rint is odd, so
    // r and the bin are the same either way): column chunks entirely below this block's rows have nothing to do
    const int i_first = blockIdx.x * kBlock;
    const int j0 = max(blockIdx.y * a.chunk, i_first + 1), j1 = min((blockIdx.y + 1) * a.chunk, a.n);
    for (int j = j0; j < j1; ++j) {
        double dx = a.x[j] - xi, dy = a.y[j] - yi, dz = a.z[j] - zi;
        dx = dx - a.L * rdf_image(dx, a.L, a.invL);
        dy = dy - a.L * rdf_image(dy, a.L, a.invL);
        dz = dz - a.L * rdf_image(dz, a.L, a.invL);
        const double r = __builtin_sqrt(dx * dx + dy * dy + dz * dz);
        if (live && j > i && r < a.rmax) {
            // int(r / dr): the product with 1/dr decides unless it lands within 1e-9 of an integer
            const double q = r * a.inv_dr;
            int bin = (int)q;
            if (q - (double)bin < 1e-9 || (double)(bin + 1) - q < 1e-9) bin = (int)(r / a.dr);
            if (bin < a.nbins) atomicAdd(&lhist[bin], 2u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < a.nbins; b += kBlock)
        if (lhist[b]) atomicAdd(&a.hist[b], (unsigned long long)lhist[b]);
}

// ===========================================================================
// Trajectory analysis, time-origin averages: the per-(origin, lag) particle means of the reference's
// compute_msd_tau_timeorig / compute_vacf_tau_timeorig (scripts/md_one_run_analysis.py:404-489):
//   MSD :  mean_i |ru_i(t0 + lag) - ru_i(t0)|^2        VACF:  mean_i v_i(t0) . v_i(t0 + lag)
// One block per (lag, origin); per element the reference's numpy expression (dx*dx + dy*dy + dz*dz with separate
// roundings; x*x0 + y*y0 + z*z0), summed over the particles in a fixed order (the reference's np.mean is a pairwise
// sum: the two agree to rounding, ~1e-16 relative, not bit for bit).  The host adds the origins in the reference's order.
// ===========================================================================
template <bool VACF>
__global__ __launch_bounds__(kBlock) void time_origin_kernel(TimeOriginArgs a)
{
    __shared__ double red[kWavesPerBlock];
    const int lag = blockIdx.x, t0 = blockIdx.y * a.origin_stride;
    if (t0 + lag >= a.n_snap || lag > min(a.max_lag, a.n_snap - 1 - t0)) return;      // (whole block: uniform)
    const size_t o0 = (size_t)t0 * a.n, o1 = (size_t)(t0 + lag) * a.n;
    double s[1] = {0.0};
    for (int i = threadIdx.x; i < a.n; i += kBlock) {
        if constexpr (VACF) {
            s[0] += a.x[o1 + i] * a.x[o0 + i] + a.y[o1 + i] * a.y[o0 + i] + a.z[o1 + i] * a.z[o0 + i];
        } else {
            const double dx = a.x[o1 + i] - a.x[o0 + i], dy = a.y[o1 + i] - a.y[o0 + i], dz = a.z[o1 + i] - a.z[o0 + i];
            s[0] += dx * dx + dy * dy + dz * dz;
        }
    }
    block_sum<1>(s, red);
    if (threadIdx.x == 0) a.term[(size_t)blockIdx.y * (a.max_lag + 1) + lag] = s[0] / (double)a.n;
}

// ---------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------
hipError_t launch_pair_rows_generic(const PairArgs &a, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL(pair_rows_generic_kernel, grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_pair_tiles(const PairArgs &a, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL(pair_tiles_kernel, grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_pair_n3(const N3Args &a, dim3 grid, int wg_waves, hipStream_t s)
{
    // 3 waves per SIMD (168 VGPRs, no spills) is the measured optimum of the register budget; wg_waves = waves
    // (= consecutive row groups) per workgroup, 2 and 4 only for 4-tile row groups
    // a.energy == 0 (forces only) exists for the one-wave workgroups; the LDS-combining forms always keep the sums
    if (a.RT == 1 && !a.energy)
        hipLaunchKernelGGL((pair_n3_kernel<3, 1, 1, false>), grid, dim3(kTile), 0, s, a);
    else if (a.RT == 1)
        hipLaunchKernelGGL((pair_n3_kernel<3, 1, 1, true>), grid, dim3(kTile), 0, s, a);
    else if (a.RT == 2 && !a.energy)
        hipLaunchKernelGGL((pair_n3_kernel<3, 2, 1, false>), grid, dim3(kTile), 0, s, a);
    else if (a.RT == 2)
        hipLaunchKernelGGL((pair_n3_kernel<3, 2, 1, true>), grid, dim3(kTile), 0, s, a);
    else if (wg_waves == 4)
        hipLaunchKernelGGL((pair_n3_kernel<3, kRowTiles, 4, true>), grid, dim3(4 * kTile), 0, s, a);
    else if (wg_waves == 2)
        hipLaunchKernelGGL((pair_n3_kernel<3, kRowTiles, 2, true>), grid, dim3(2 * kTile), 0, s, a);
    else if (!a.energy)
        hipLaunchKernelGGL((pair_n3_kernel<3, kRowTiles, 1, false>), grid, dim3(kTile), 0, s, a);
    else
        hipLaunchKernelGGL((pair_n3_kernel<3, kRowTiles, 1, true>), grid, dim3(kTile), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_pair_n3_f32(const N3Args &a, dim3 grid, hipStream_t s)
{
    if (a.energy)
        hipLaunchKernelGGL(pair_n3_f32_kernel<true>, grid, dim3(kTile), 0, s, a);
    else
        hipLaunchKernelGGL(pair_n3_f32_kernel<false>, grid, dim3(kTile), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_tile_boxes(const GeometryArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(tile_boxes_kernel, dim3((a.T + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_tile_mask(const GeometryArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(tile_mask_kernel, dim3(a.W, (a.TB + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_tile_class(const GeometryArgs &a, double invL, double rc2, int S, int NGo, unsigned *desc, unsigned *desc_far,
                             float *desc2, hipStream_t s)
{
    const dim3 grid((a.T + kBlock - 1) / kBlock, NGo);
    if (a.RT == 1)
        hipLaunchKernelGGL(tile_class_kernel<1>, grid, dim3(kBlock), 0, s, a, invL, rc2, S, desc, desc_far, desc2);
    else if (a.RT == 2)
        hipLaunchKernelGGL(tile_class_kernel<2>, grid, dim3(kBlock), 0, s, a, invL, rc2, S, desc, desc_far, desc2);
    else
        hipLaunchKernelGGL(tile_class_kernel<kRowTiles>, grid, dim3(kBlock), 0, s, a, invL, rc2, S, desc, desc_far, desc2);
    return hipGetLastError();
}

hipError_t launch_drift_kick(const IntegrateArgs &a, int phase, hipStream_t s)
{
    const dim3 grid((a.rows + kBlock - 1) / kBlock);
    if (phase == 1)
        hipLaunchKernelGGL(drift_kick_kernel<1>, grid, dim3(kBlock), 0, s, a);
    else if (phase == 2)
        hipLaunchKernelGGL(drift_kick_kernel<2>, grid, dim3(kBlock), 0, s, a);
    else if (a.bbox)
        hipLaunchKernelGGL((drift_kick_kernel<0, true>), grid, dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL(drift_kick_kernel<0>, grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_reduce_forces(const ReduceArgs &a, bool all_blocks, hipStream_t s)
{
    const dim3 grid(a.P / kTile, all_blocks ? a.G : 1);
    if (a.slab_j)
        hipLaunchKernelGGL(reduce_forces_kernel<true>, grid, dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL(reduce_forces_kernel<false>, grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_kick(const IntegrateArgs &a, bool kick, hipStream_t s)
{
    const dim3 grid((a.rows + kBlock - 1) / kBlock);
    if (kick)
        hipLaunchKernelGGL(kick_kernel<true>, grid, dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL(kick_kernel<false>, grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_kick_finalize(const IntegrateArgs &a, const FinalizeArgs &f, bool kick, hipStream_t s)
{
    const dim3 grid((a.rows + kBlock - 1) / kBlock);
    if (kick)
        hipLaunchKernelGGL(kick_finalize_kernel<true>, grid, dim3(kBlock), 0, s, a, f);
    else
        hipLaunchKernelGGL(kick_finalize_kernel<false>, grid, dim3(kBlock), 0, s, a, f);
    return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void sum_blocks_kernel(const double *blocks, double *out, int G, int len)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= len) return;
    double t = blocks[i];
    for (int g = 1; g < G; ++g) t += blocks[(size_t)g * len + i];
    out[i] = t;
}

hipError_t launch_sum_blocks(const double *blocks, double *out, int G, int len, hipStream_t s)
{
    hipLaunchKernelGGL(sum_blocks_kernel, dim3((len + kBlock - 1) / kBlock), dim3(kBlock), 0, s, blocks, out, G, len);
    return hipGetLastError();
}

hipError_t launch_kinetic_fused(const IntegrateArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(kinetic_fused_kernel, dim3((a.rows + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_time_origin(const TimeOriginArgs &a, bool vacf, int n_origins, hipStream_t s)
{
    const dim3 grid(a.max_lag + 1, n_origins);
    if (vacf)
        hipLaunchKernelGGL(time_origin_kernel<true>, grid, dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL(time_origin_kernel<false>, grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_rdf_histogram(const RdfArgs &a, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL(rdf_histogram_kernel, grid, dim3(kBlock), (size_t)a.nbins * sizeof(unsigned), s, a);
    return hipGetLastError();
}

hipError_t launch_tile_tail(const ReduceArgs &ra, const IntegrateArgs &a, const FinalizeArgs &f, const FinalizeArgs &prev, bool kick,
                            bool drift, hipStream_t s)
{
    // a.ticket == NULL: the record of this launch is folded by the next one; prev.ring != NULL: a record is pending --
    // it gets an extra block here, or (with tickets) the last block takes it first
    // one block per Newton-3 row group of 1 or 2 tiles (the gather kernel's tiles stand alone)
    const bool n3 = ra.slab_j != nullptr;
    const int tpb = (n3 && ra.RT == 2) ? 2 : 1;
    const dim3 grid(a.P / (kTile * tpb) + ((!a.ticket && prev.ring) ? 1 : 0)), block(kBlock * tpb);
#define LJMD_TAIL(N3_, KICK_, DRIFT_, TPB_) hipLaunchKernelGGL((tile_tail_kernel<N3_, KICK_, DRIFT_, TPB_>), grid, block, 0, s, ra, a, f, prev)
    if (n3 && tpb == 2) {
        if (kick && drift) LJMD_TAIL(true, true, true, 2);
        else if (kick) LJMD_TAIL(true, true, false, 2);
        else LJMD_TAIL(true, false, false, 2);
    } else if (n3) {
        if (kick && drift) LJMD_TAIL(true, true, true, 1);
        else if (kick) LJMD_TAIL(true, true, false, 1);
        else LJMD_TAIL(true, false, false, 1);
    } else {
        if (kick && drift) LJMD_TAIL(false, true, true, 1);
        else if (kick) LJMD_TAIL(false, true, false, 1);
        else LJMD_TAIL(false, false, false, 1);
    }
#undef LJMD_TAIL
    return hipGetLastError();
}

hipError_t launch_finalize(const FinalizeArgs &a_in, double *fold_scratch, hipStream_t s)
{
    FinalizeArgs a = a_in;
    if (a.n_wg > kDirectFoldMax && fold_scratch) {
        hipLaunchKernelGGL(fold_partials_kernel, dim3(kFoldBlocks), dim3(kBlock), 0, s, a.wg_part, fold_scratch, a.n_wg);
        a.wg_part = fold_scratch;
        a.n_wg = kFoldBlocks;
    }
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace ljmdk

#ifdef LJMD_WAVE_TRACE
extern "C" int ljmd_debug_wave_trace(unsigned long long *out, int n_waves)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ljmdk::g_wave_trace), (size_t)n_waves * 8 * sizeof(unsigned long long));
}
#endif
#ifdef LJMD_VARIANT_STATS
extern "C" int ljmd_debug_variant_stats(unsigned long long *out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(ljmdk::g_variant_stats), 80 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[80] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(ljmdk::g_variant_stats), z, sizeof z);
    }
    return (int)e;
}
#endif
