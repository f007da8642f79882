// ljmd_internal.h -- argument blocks shared by the kernels and the C-ABI host code.
//
// HBM layout (all fp64, structure of arrays):
//   n      total particles, G ranks, S = n / G particles per rank ("shard"),
//   P      axis stride = S rounded up to a multiple of kBlock (256); slots S..P-1 of every
//          axis array are padding: position = NaN (never passes r^2 < rc^2), v = a = 0.
//   pos    exchange buffer [G][3][P]: block g = x[P] y[P] z[P] of rank g's particles.
//   ru,v,a [3][P] of the owned shard.
//   tiles  64 consecutive slots; TB = P / 64 tiles per rank block, T = G * TB tiles.
#ifndef LJMD_INTERNAL_H
#define LJMD_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ljmdk {

constexpr int kBlock = 256;                 // threads per workgroup = 4 wave64
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kTile = 64;                   // particles per tile = one wave
#ifndef LJMD_ROW_TILES
#define LJMD_ROW_TILES 4
#endif
constexpr int kRowTiles = LJMD_ROW_TILES;   // tiles per Newton-3 row group (particles per lane)
constexpr unsigned kAllRows = (1u << kRowTiles) - 1u;
constexpr int kSlotAlign = (kRowTiles * 64 > 256) ? kRowTiles * 64 : 256;   // P is a multiple of this
constexpr int kPartialStride = 8;           // doubles per per-rank per-step partial record
constexpr int kFoldBlocks = 128;             // pre-reduction blocks of the scalar partials (large grids)
constexpr int kDirectFoldMax = 20480;        // workgroup partials up to which ONE block folds the step record (finalize_body) in every launch form
constexpr int kBoxStride = 8;               // doubles per tile bounding box (lo xyz, hi xyz, 2 pad)

struct PairArgs {
    const double *pos;      // exchange buffer [G][3][P]
    double *slab;           // [nslab][3][P] raw partial accelerations of the owned rows
    double *wg_part;        // [n workgroups][2] = s12, s6
    const uint64_t *mask;   // [TB][W] bit (J) of row I set = tile pair must be evaluated
    const double *bbox;     // tile kernel with inline_mask: the tile boxes [T][kBoxStride]
    int inline_mask;        // tile kernel, small single-rank systems: the waves work their mask words out themselves
                            // (tile_mask_kernel's test; `mask` unused) and the step saves a launch
    double rc2_skin;        // rc^2 (1 + 1e-10) of that test
    int n;                  // total particles
    int S;                  // real particles per rank
    int P;                  // padded axis stride
    int G;                  // ranks
    int rank;
    int TB;                 // tiles per rank block
    int T;                  // tiles in total
    int W;                  // 64-bit mask words per row = ceil(T / 64)
    int chunk;              // generic kernel: j slots per grid.y slice; tile kernel: column tiles per slice
    double L, invL, rc2;
};

// Newton-3 pair kernel (ljmd_kernels.hip: pair_n3_kernel).  Row group = kRowTiles (4) consecutive
// tiles = 256 slots held by ONE wave, 4 particles per lane.  NG = T / 4 groups.  A row group A
// owns the column groups B with cyclic offset d = (B - A) mod NG in [0, Dmax] (ties broken by
// index), so every unordered pair of particles is evaluated exactly once on the whole machine.
struct N3Args {
    const double *pos;      // exchange buffer
    const uint64_t *mask;   // [TB rows][W] tile-pair mask of the owned row tiles
    const double *bbox;     // [T][kBoxStride] exact tile bounding boxes (fp32 far kernel: image classification)
    const unsigned *desc;   // [NGo][T] pass descriptors of tile_class_kernel (row-tile mask bits, loop variant, images)
    const float *desc2;     // [NGo][T][8] cluster passes: direction n (3), thresholds of the RT row tiles (4), pad; or NULL
    double *slab_i;         // [nchunk][3][P] partial accelerations of the owned rows (row side)
    double *slab_j;         // [T][CS][3][64] column-side partial accelerations: one block per (column tile, workgroup of WG
                            // consecutive row groups); the CS blocks of a column tile lie together, so that the reduction
                            // streams them.  Block j of column tile c: by_group ? the workgroup's index : e / WG, with e =
                            // the column group's offset from the workgroup's first row group (single rank, WG | NG: one
                            // workgroup per value, CS = (Dmax + WG - 1) / WG + 1)
    unsigned char *flag_j;  // [T][CS] 1 = slab_j block written this step
    double *wg_part;        // [n workgroups][2]
    int S, P, G, rank, TB, T, W;   // S = real particles per rank (slots S..P-1 are padding)
    int NG, NGo, Dmax;      // row groups in total / owned by this rank (NGo = TB / RT, NG = G * NGo)
    int CS, by_group;       // slab_j layout (above)
    int RT;                 // tiles per row group: 4 for large systems, 1 or 2 to give small ones enough work items
    int dchunk;             // fp32 far kernel: offsets d per grid.y slice
    int uchunk;             // pair_n3_kernel: UNITS per grid.y slice.  Unit u of a row group = e * RT + l: offset e, column tile l of
                            // the group there (one pass); = dchunk * RT unless the system is cut finer
    int energy;             // 0: forces only -- the energy sums are not accumulated and the workgroup partials are NaN
    int xcd_remap;          // C > 0: XCD-aware mapping, chunks of C consecutive row groups per XCD (gridDim.x % (8 C) == 0)
    int inline_class;       // RT <= 2, one wave per workgroup: the waves compute their pass descriptors themselves (desc unused)
    int both_ties;          // one wave per workgroup, one rank: the tie d = NG / 2 (NG even) is worked from both sides, the
                            // steps 0 .. 31 by the lower row group and 1 .. 32 by the upper one (ljmd_kernels.hip: tile_of)
    double L, invL, rc2;
    double rc2_skin;        // rc^2 (1 + 1e-10) of the tile-pair test (GeometryArgs::rc2_skin), for inline_class
};

struct IntegrateArgs {
    double *r;              // own block of the exchange buffer, axis stride = P
    double *ru, *v, *a;     // [3][P]
    const double *fsum;     // [3][P] raw (prefactor-free) total accelerations of the owned rows
    double *ke_part;        // [n blocks][3]
    int rows;               // = P (padding included: it integrates to NaN / 0 harmlessly)
    int P;
    double L, invL, dt, dt_half, dt_sq_half;
    double *bbox;           // drift kernel, single rank: also emit the tile bounding boxes [P / 64][kBoxStride] (else NULL)
    double *pos_tc;         // ... and, on the Newton-3 path, the tile-coherent copy of the new positions [3][P] (else NULL)
    int RT;                 // tiles per Newton-3 row group (the frame of pos_tc is per group)
    unsigned *ticket;       // kick kernel with the finalize folded in: blocks-done counter (else NULL)
};

// Deterministic reduction of the pair kernels' partial-acceleration slabs into fpart.
// Gather kernels / single rank: fpart = [1][3][P] (own rows).  Newton-3 on G > 1 ranks:
// fpart = [G][3][P], block g = this rank's contributions to rank g's particles (own block:
// row side + column side, remote blocks: column side only); an RCCL reduce-scatter then sums
// block g over the ranks into rank g's fsum.
struct ReduceArgs {
    const double *slab;     // row side [nslab][3][P]
    const double *slab_j;   // column side (Newton-3) or NULL
    const unsigned char *flag_j;
    const float *slab_j2;   // second column-side set (fp32 far pass of the mixed-precision mode: [T][CS2] blocks of floats) or NULL
    const unsigned char *flag_j2;
    double *fpart;
    int nslab, P, G, rank, TB;
    int CS, CS2;            // blocks per column tile of slab_j / slab_j2 (N3Args::slab_j)
    int RT;                 // tiles per Newton-3 row group of this engine (1, 2 or 4)
};

struct FinalizeArgs {
    const double *wg_part;
    const double *ke_part;
    const double *ke_tile;  // NULL, or [T][3] per-tile sums of v^2 (tile_tail_kernel) to be combined per 256-slot block
    double *ring;           // [ring_cap][kPartialStride]
    unsigned *ring_pos;     // device counter, bumped once per finalize
    int n_wg, n_ke;
    unsigned ring_cap;
    double pair_scale;      // 0.5 when every unordered pair was visited twice (gather kernels), 1 for Newton-3
};

struct GeometryArgs {
    const double *pos;      // exchange buffer
    double *bbox;           // [T][kBoxStride]
    double *pos_tc;         // Newton-3 path: tile-coherent copy of pos, same layout (else NULL): every tile's particles in the
                            // periodic image nearest to the tile's first particle; bbox then bounds THOSE coordinates
    uint64_t *mask;         // [TB][W]
    uint64_t *mask_far;     // mixed precision only (else NULL): [TB][W] tile pairs evaluated in fp32;
                            // `mask` then holds only the NEAR pairs (box distance <= r_split, or same row group)
    int P, G, rank, TB, T, W;
    int RT;                 // tiles per Newton-3 row group (same-group pairs always count as NEAR)
    double L, invL, rc2_skin;   // rc^2 * (1 + 1e-10): skip only when provably outside
    double rsplit2;         // r_split^2
    double rvfar2;          // mixed precision: box distance^2 beyond which a far pass is VERY FAR (u^3 = r^-6 < 2^-26: the u^6
                            // terms lie below the fp32 resolution of the u^3 terms; pair_n3_f32<., VFAR>), or +inf: never
    int pertile_images;     // tile_class: row tiles may take their own periodic image on a single general axis (LJMD_N3_PERTILE)
    int both_ties;          // tile_class: the tie d = NG / 2 is visited from both sides (N3Args::both_ties)
};

struct RdfArgs {
    const double *x, *y, *z;       // [n] one snapshot, wrapped coordinates
    unsigned long long *hist;      // [nbins] ordered-pair counts (added to)
    int n, nbins, chunk;           // chunk = j per grid.y slice
    double L, rmax, dr;
    double invL, inv_dr;           // 1 / L, 1 / dr: fast paths of the two divisions (exact path kept for near-ties)
};

struct TimeOriginArgs {
    const double *x, *y, *z;       // [n_snap][n] unwrapped positions (MSD) or velocities (VACF)
    double *term;                  // [n_origins][max_lag + 1] particle mean per (origin, lag); entries beyond an origin's reach untouched
    int n_snap, n, max_lag, origin_stride;
};

hipError_t launch_pair_rows_generic(const PairArgs &a, dim3 grid, hipStream_t s);
hipError_t launch_pair_tiles(const PairArgs &a, dim3 grid, hipStream_t s);
hipError_t launch_pair_n3(const N3Args &a, dim3 grid, int wg_waves, hipStream_t s);   // dispatches on a.RT, wg_waves
hipError_t launch_pair_n3_f32(const N3Args &a, dim3 grid, hipStream_t s);
hipError_t launch_drift_kick(const IntegrateArgs &a, int phase /* 0 all, 1 positions, 2 velocities */, hipStream_t s);
hipError_t launch_reduce_forces(const ReduceArgs &a, bool all_blocks, hipStream_t s);
hipError_t launch_kick(const IntegrateArgs &a, bool kick, hipStream_t s);
// kick + finalize in one launch: the last block to finish folds the partials (needs a.ticket, f.n_wg <= kDirectFoldMax)
hipError_t launch_kick_finalize(const IntegrateArgs &a, const FinalizeArgs &f, bool kick, hipStream_t s);
hipError_t launch_kinetic_fused(const IntegrateArgs &a, hipStream_t s);
// out[i] = blocks[0][i] + blocks[1][i] + ... + blocks[G-1][i], i < len (left to right: the order is part of the result)
hipError_t launch_sum_blocks(const double *blocks, double *out, int G, int len, hipStream_t s);
hipError_t launch_finalize(const FinalizeArgs &a, double *fold_scratch /* [2 * kFoldBlocks] or NULL */, hipStream_t s);
// small single-rank systems: slab reduction + kick + step record (+ the next step's K1 when drift) in one launch, one
// block per tile; kick without drift = the last step of a batch, neither = a plain force evaluation
hipError_t launch_tile_tail(const ReduceArgs &ra, const IntegrateArgs &a, const FinalizeArgs &f, const FinalizeArgs &prev, bool kick,
                            bool drift, hipStream_t s);
hipError_t launch_rdf_histogram(const RdfArgs &a, dim3 grid, hipStream_t s);
hipError_t launch_time_origin(const TimeOriginArgs &a, bool vacf, int n_origins, hipStream_t s);
hipError_t launch_tile_boxes(const GeometryArgs &a, hipStream_t s);
hipError_t launch_tile_mask(const GeometryArgs &a, hipStream_t s);
hipError_t launch_tile_class(const GeometryArgs &a, double invL, double rc2, int S, int NGo, unsigned *desc,
                             unsigned *desc_far, float *desc2 /* cluster passes, or NULL: none */, hipStream_t s);

// ljmd_sort.hip
size_t kd_temp_bytes(int count);
hipError_t launch_iota(int *idx, int P, hipStream_t s);
hipError_t kd_level(void *temp, size_t temp_bytes, const double *coord_axis, double L, unsigned long long *keys,
                    unsigned long long *keys_out, int *idx, int *idx_out, int S, int nseg, const int *seg_offsets,
                    hipStream_t s);
// ownership migration of a multi-GPU run (ljmd_sort.hip; ljmd_capi.cpp: migrate_pack / migrate_deal)
constexpr int kMigrateRows = 10;            // doubles per slot in the migration buffer: ru, v, a (3 each) + the particle id
hipError_t launch_iota_blocked(int *idx, int n, int S, int P, hipStream_t s);
hipError_t launch_iota_offset(int *idx, int count, int P, int offset, hipStream_t s);
hipError_t kd_level_blocked(void *temp, size_t temp_bytes, const double *pos, int axis, int P, double L,
                            unsigned long long *keys, unsigned long long *keys_out, int *idx, int *idx_out, int n, int nseg,
                            const int *seg_offsets, hipStream_t s);
hipError_t launch_migrate_pack(const double *ru, const double *v, const double *a, const int *perm, const int *gid0,
                               double *block, int S, int P, hipStream_t s);
hipError_t launch_migrate_select(const double *pos_all, const double *mig_all, const int *mine, double *new_pos, double *ru,
                                 double *v, double *a, int *gid0, int S, int P, hipStream_t s);
hipError_t launch_gather3(const double *src, double *dst, const int *idx, int P, hipStream_t s);
hipError_t launch_gather_perm(const int *src, int *dst, const int *idx, int P, hipStream_t s);

}  // namespace ljmdk
#endif
