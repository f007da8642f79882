// ljmd_internal.h -- argument blocks shared by the kernels and the C-ABI host code.
#ifndef LJMD_INTERNAL_H
#define LJMD_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ljmdk {

constexpr int kBlock = 256;                 // threads per workgroup = 4 wave64
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kPartialStride = 8;           // doubles per per-rank per-step partial record

struct PairArgs {
    const double *pos;   // exchange buffer: all n positions, shard-blocked SoA
    double *slab;        // [nslab][3][shard] raw partial accelerations of the owned rows
    double *wg_part;     // [n workgroups][2] = s12, s6
    int n;               // total particles
    int shard;           // particles per rank (n / n_ranks)
    int rank;
    int row0;            // global index of the first owned row (= rank * shard)
    int rows;            // owned rows (= shard)
    int chunk;           // j per grid.y slice
    double L, invL, rc2;
};

struct IntegrateArgs {
    double *r;           // own block of the exchange buffer, axis stride = shard
    double *ru, *v, *a;  // [3][shard]
    const double *slab;  // [nslab][3][shard]
    double *ke_part;     // [n blocks][3]
    int rows, shard, nslab;
    double L, invL, dt, dt_half, dt_sq_half;
};

struct FinalizeArgs {
    const double *wg_part;
    const double *ke_part;
    double *ring;        // [ring_cap][kPartialStride]
    unsigned *ring_pos;  // device counter, bumped once per finalize
    int n_wg, n_ke;
    unsigned ring_cap;
};

hipError_t launch_pair_rows(const PairArgs &a, bool fast_mic, dim3 grid, hipStream_t s);
hipError_t launch_drift_kick(const IntegrateArgs &a, hipStream_t s);
hipError_t launch_reduce_kick(const IntegrateArgs &a, bool kick, hipStream_t s);
hipError_t launch_kinetic_fused(const IntegrateArgs &a, hipStream_t s);
hipError_t launch_finalize(const FinalizeArgs &a, hipStream_t s);

}  // namespace ljmdk
#endif
