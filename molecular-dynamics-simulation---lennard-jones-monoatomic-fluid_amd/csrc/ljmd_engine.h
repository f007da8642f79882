// ljmd_engine.h -- the engine object behind the opaque ljmd_t of include/ljmd.h and the internal
// step-phase functions shared by ljmd_capi.cpp (C ABI) and ljmd_multi.cpp (single-process multi-device).
#ifndef LJMD_ENGINE_H
#define LJMD_ENGINE_H

#include "ljmd.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "ljmd_internal.h"

using namespace ljmdk;

namespace ljmdh {

extern thread_local std::string g_last_error;

constexpr unsigned kRingCap = 4096;     // per-step partial records kept on the device
constexpr int kTargetWorkgroups = 2048; // >> 256 CUs (8 per CU) for the pair kernels
constexpr int kXcdMinGroups = 256;      // row groups per rank from which the XCD-aware work mapping is used (neutral below)
constexpr int kMixedMinN = 16384;       // smallest system the mixed-precision mode accepts
constexpr long kN3ItemsFor4 = 6000;     // (row group, offset) pairs a rank needs before 4 (n >= 32768 on one rank) ...
constexpr long kN3ItemsFor2 = 1000;     // ... or 2 (n >= 6144) tiles per row group pay off: profiles/r04_unit_sweep.txt
constexpr long kN3LargeItems = 131072;  // (row group, offset) pairs of a rank from which work items are slices of whole offsets
constexpr int kN3MidTargetItems = 32768;   // work items aimed at below that on one rank (units of one pass)
constexpr int kBothTiesMaxGroups = 128; // row groups up to which the tie d = NG / 2 is worked from both sides
constexpr int kFuseTailMaxRowTiles = 2; // tiles per row group up to which small systems take the two-launch step ...
constexpr int kFuseTailMaxN = 20000;    // ... and their largest particle count (n = 24 576: the in-kernel pass descriptors cost the pair kernel more than the launches save)
constexpr int kMaxProfiledLaunches = 4096;
constexpr int kEventsPerLaunch = 9;

// md_types.f90:22
constexpr double kPi = 3.1415926535897932384626433832795;

struct EventSet {
    // 0: before K1, 1: before geometry, 2: before pair, 3: after pair, 4: end (engine stream);
    // 5 / 6: around the position exchange, 7 / 8: around the force exchange, on the stream that carries them
    hipEvent_t e[kEventsPerLaunch];
    bool has_pos_x = false, has_force_x = false;   // events 5 / 6 and 7 / 8 were recorded for this launch
};

inline int env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : dflt;
}

}  // namespace ljmdh

struct ljmd_multi;

using namespace ljmdh;

struct ljmd {
    // single-process multi-device parent (ljmd_create_multi, ljmd_multi.cpp): owns one child engine per device
    // and no device memory of its own; every public entry point dispatches on it
    ljmd_multi *multi = nullptr;
    bool poisoned = false;            // a batch of steps failed half-way: LJMD_ERR_STATE until ljmd_set_state
    int inject_failure_at = -1;       // LJMD_INJECT_FAILURE_AT_STEP (fault injection for the tests): the force phase of the
                                      // evaluation with this index fails once, behind an already enqueued drift
    // ---- parameters (type(sim_params), md_types.f90:27-50) ----
    int n = 0, S = 0, P = 0, rank = 0, G = 1, device = 0, mode = 0;
    int TB = 0, T = 0, W = 0;
    double L = 0, invL = 0, volume = 0, rc = 0, rc2 = 0, dt = 0, dt_half = 0, dt_sq_half = 0;
    double tail_e = 0, tail_d = 0, tail_dd = 0;
    bool tail_on = true;              // ljmd_set_tail_corrections (the reference's use_tail_corrections, default .true.)
    bool rc_allows_fast = false;      // rc <= (1 - 1e-9) * L/2
    bool positions_compact = false;   // coordinate spread < 2.4 L (always true after a wrap)
    bool have_state = false, have_accel = false;
    bool sort_enabled = true;
    bool force_generic = false;       // LJMD_FORCE_GENERIC=1: always take the exact generic kernel (A/B tests)
    bool force_collectives = false;   // LJMD_FORCE_COLLECTIVES=1: a 1-rank engine still issues its RCCL calls (tests)
    int resort_every = 20, steps_since_sort = 0;

    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;        // RCCL communicator over the G ranks (multi-GPU only)
    // position all-gather overlapped with the velocity half-kick: the collective runs on comm_stream between
    // ev_pos_ready (positions drifted, engine stream) and ev_gather_done (awaited by the engine stream)
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_pos_ready = nullptr, ev_gather_done = nullptr;
    bool overlap_exchange = true;     // LJMD_OVERLAP_EXCHANGE
    bool gather_done_for_step = false;
    // ---- HBM-resident state (layout: ljmd_internal.h) ----
    double *d_pos = nullptr;      // [G][3][P] exchange buffer (all positions)
    double *d_ru = nullptr, *d_v = nullptr, *d_a = nullptr;   // [3][P]
    double *d_slab = nullptr;     // [nslab_max][3][P]
    double *d_wg_part = nullptr;  // [n_wg_max][2]
    double *d_fold = nullptr;     // [kFoldBlocks][2]
    unsigned *d_ticket = nullptr; // blocks-done counter of the kick kernel with the finalize folded in
    bool want_energy = true;      // false: the next force evaluations skip the energy sums (epot, d_epot, dd_epot = NaN)
    int xcd_remap = 0;            // LJMD_N3_XCD_REMAP: consecutive row groups per XCD chunk of the Newton-3 pair kernel (0 = plain mapping)
    bool fuse_small = true;       // LJMD_FUSE: boxes inside the drift kernel, finalize inside the kick kernel
    // LJMD_FUSE_TAIL (default on): small single-rank systems run a step as TWO launches -- the pair kernel with its pass
    // descriptors worked out in-kernel, and tile_tail_kernel (slab reduction + kick + step record + the NEXT step's K1).
    // kick_hint / next_drift_hint: what the caller of enqueue_pair_forces already knows about the rest of the step (-1 = not
    // known: the split-phase API); reduce_deferred: the pair phase left the reduction to the tail launch of enqueue_kick;
    // drift_prefused: the previous step's tail has already run this step's K1
    bool fuse_tail = false;
    int kick_hint = -1;
    bool next_drift_hint = false;
    bool reduce_deferred = false, deferred_n3 = false;
    int deferred_nslab = 0;
    bool drift_prefused = false;
    double *d_ke_tile = nullptr;      // [T][3] per-tile sums of v^2
    bool boxes_valid = false;     // d_bbox already holds the boxes of the current positions (written by the drift kernel)
    double *d_ke_part = nullptr;  // [n_ke][3]
    double *d_ring = nullptr;     // [kRingCap][kPartialStride]
    unsigned *d_ring_pos = nullptr;
    double *d_bbox = nullptr;     // [T][kBoxStride]
    uint64_t *d_mask = nullptr;   // [TB][W]
    double *d_pos_tc = nullptr;   // [G][3][P] tile-coherent copy of d_pos for the Newton-3 kernels (tile_boxes_kernel)
    unsigned *d_desc = nullptr;   // [NGo][T] pass descriptors of the Newton-3 kernel (tile_class_kernel)
    unsigned *d_desc_far = nullptr;   // same for the fp32 far kernel of the mixed mode (from mask_far)
    float *d_desc2 = nullptr;     // [NGo][T][8] direction + thresholds of the cluster passes (LJMD_N3_CLUSTERS, default on)
    // sorting scratch
    int *d_idx = nullptr, *d_idx2 = nullptr, *d_perm = nullptr, *d_perm2 = nullptr;
    double *d_tmp3 = nullptr;     // [3][P]
    void *d_cub = nullptr;
    size_t cub_bytes = 0;
    // k-d ordering (default): per level the segment boundaries in particle units
    std::vector<int> kd_level_nseg;       // segments at level l
    std::vector<size_t> kd_level_off;     // offset of level l's boundaries inside d_kd_offsets
    std::vector<int> kd_axis;             // split axis of level l: always the longest remaining extent
    int *d_kd_offsets = nullptr;
    unsigned long long *d_kd_keys = nullptr, *d_kd_keys2 = nullptr;   // [P]

    unsigned ring_consumed = 0;   // host mirror: records already read back
    unsigned ring_issued = 0;     // host mirror: finalize launches issued
    // launch geometry
    int nslab_g = 1, chunk_g = 0;     // generic kernel: grid (P/256, nslab_g), chunk_g j per slice
    int nslab_t = 1, chunk_t = 0;     // tile kernel:    grid (TB/4, nslab_t), chunk_t column tiles per slice
    // Newton-3 kernel (single rank): NG row groups, offsets 0..Dmax in nslab_n slices of dchunk
    bool use_n3 = false;
    int wg_waves = 1;                 // LJMD_N3_WG_WAVES: row groups (waves) per pair-kernel workgroup (1, 2, 4)
    bool both_ties = false;           // one rank, one wave per workgroup: the tie d = NG / 2 is worked from both sides (N3Args::both_ties)
    int uchunk = 0;                   // units (passes) per work item (N3Args::uchunk)
    // two launches per step, record fold off the critical path (tile_tail_kernel): the step's record is folded by the NEXT
    // tail launch of the batch; the workgroup partials and per-tile v^2 sums it reads alternate between two buffers
    bool fold_pending = false, defer_record = true;
    ljmdk::FinalizeArgs pending_fold{};
    int fold_parity = 0;
    size_t wg_part_stride = 0;        // doubles per wg_part buffer
    int CS = 0, CS2 = 0, j_by_group = 0;   // slab_j / slab_j2: blocks per column tile, block numbering (N3Args::slab_j)
    int NG = 0, NGo = 0, Dmax = 0, nslab_n = 1, dchunk = 0;
    int rt = kRowTiles;               // tiles per row group (LJMD_N3_ROW_TILES; auto: 4, or 2 / 1 for small systems)
    double *d_slab_j = nullptr;
    unsigned char *d_flag_j = nullptr;
    // mixed precision (mode = LJMD_PRECISION_FP32_FORCE): far tile pairs in fp32
    uint64_t *d_mask_far = nullptr;
    double *d_slab_j2 = nullptr;
    unsigned char *d_flag_j2 = nullptr;
    double r_split = 5.0;             // LJMD_FP32_SPLIT: boxes closer than this stay fp64
    // reduced raw accelerations: fpart [G or 1][3][P]; frecv [3][P] = reduce-scatter result (G > 1, Newton-3)
    double *d_fpart = nullptr, *d_frecv = nullptr;
    double *d_fall = nullptr;         // [G][3][P] blocks received in the all-to-all form of the force exchange
    bool exchange_alltoall = false;   // LJMD_FORCE_EXCHANGE=alltoall: direct sends + local rank-order sum
    bool forces_pending = false;      // pair kernel + slab reduction enqueued, kick not yet
    bool external_force_exchange = false;   // tests: the caller sums fpart over ranks into frecv
    int pending_n_wg = 0;
    double pending_scale = 0.5;
    int n_ke = 0;

    double *h_stage = nullptr;    // pinned, 3*G*P doubles
    double *h_ring = nullptr;     // pinned, kRingCap records
    std::vector<int> h_perm;      // slot -> original local index (>= S on padding)
    bool perm_dirty = false;

    // asynchronous snapshot (ljmd_snapshot_begin/end): device copy of r, ru, v, a [4][3][P] + perm [P],
    // its pinned host mirror, the second stream that carries the HBM -> host transfer
    // mixed precision: the fp32 far pass runs on its own stream beside the fp64 near pass (LJMD_FP32_FAR_STREAM, default on)
    hipStream_t far_stream = nullptr;
    hipEvent_t ev_far_go = nullptr, ev_far_done = nullptr;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_snap_ready = nullptr, ev_snap_done = nullptr;
    double *d_snap = nullptr, *h_snap = nullptr;
    int *d_snap_perm = nullptr, *h_snap_perm = nullptr;
    bool snap_in_flight = false;
    bool snap_ready = false;      // all of the above exist

    // Ownership migration (multi-GPU, ljmd_capi.cpp: migrate_pack / migrate_deal).  gid0[j] = particle id -- global index in
    // the order of the last ljmd_set_state (or of the last rebase) -- of the particle at position j of this engine's arrays
    // (its "original order": what get_state returns and the slot permutation refers to); identity (rank S + j) until a deal
    int *d_gid0 = nullptr;            // [P]
    std::vector<int> h_gid0;          // host mirror, valid while `migrated`
    bool migrated = false;            // the shard is no longer the index range [rank S, (rank + 1) S) of the caller's arrays
    double *d_mig = nullptr;          // [G][kMigrateRows][P] migration buffer (lazy)
    int *d_mig_idx = nullptr, *d_mig_idx2 = nullptr;                    // [n] storage indices g P + s
    unsigned long long *d_mig_keys = nullptr, *d_mig_keys2 = nullptr;   // [n]
    int *d_mig_offsets = nullptr;
    void *d_mig_cub = nullptr;
    size_t mig_cub_bytes = 0;
    std::vector<int> mig_level_nseg, mig_axis;
    std::vector<size_t> mig_level_off;
    double mig_ext[3] = {0, 0, 0};    // extents of a rank's block after the deal (choose the shard's k-d axes)
    int32_t migrations = 0;

    bool profiling = false;
    std::vector<EventSet> ev_pool;
    size_t ev_used = 0;

    std::string err;
};


namespace ljmdh {

int fail(const ljmd_t *h, int code, const char *fmt, ...);

#define LJMD_HIP(h, call)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return ljmdh::fail((h), LJMD_ERR_HIP, "%s failed: %s (%s:%d)", #call,           \
                               hipGetErrorString(e_), __FILE__, __LINE__);                  \
    } while (0)

inline double *own_block(ljmd_t *h) { return h->d_pos + (size_t)h->rank * 3 * h->P; }
inline bool needs_force_exchange(const ljmd_t *h) { return h->use_n3 && (h->G > 1 || h->force_collectives); }

// one MD step = enqueue_drift | position exchange (multi-rank) | enqueue_pair_forces | force exchange | enqueue_kick
int enqueue_drift(ljmd_t *h, EventSet *q);
// the same in two halves, for a caller that issues the position exchange itself between them (ljmd_multi.cpp):
// positions (+ re-sort on a re-sort step; *split = false then, and the velocity half-kick has already run) ...
int enqueue_drift_positions(ljmd_t *h, EventSet *q, bool *split);
// ... and the velocity half-kick, concurrent with the exchange
int enqueue_drift_velocities(ljmd_t *h);
int enqueue_pair_forces(ljmd_t *h, EventSet *q);
int enqueue_kick(ljmd_t *h, bool kick, EventSet *q);
// both phases from one caller; next_drift: another step follows in the same batch (its K1 may ride on this step's tail launch)
int enqueue_forces(ljmd_t *h, bool kick, EventSet *q, bool next_drift = false);
int fetch_ring(ljmd_t *h, unsigned count);
void combine_one(const ljmd_t *h, const double *recs, int n_ranks, double *epot, double *ekin, double *d_epot,
                 double *dd_epot);
EventSet *next_events(ljmd_t *h);
void release(ljmd_t *h);
// ownership migration in three phases, all on the engine's stream: (1) the own block of the migration buffer <- ru, v, a,
// ids; (2) -- by the caller or the library -- all-gather of the G blocks (kMigrateRows * P doubles each; block g at
// migrate_buffer(h) + g * kMigrateRows * P); (3) the deal + the rank's new state (the exchange buffer must hold everybody's
// CURRENT positions).  Afterwards the position exchange has to run again.
int migrate_pack(ljmd_t *h);
double *migrate_buffer(ljmd_t *h);
int migrate_deal(ljmd_t *h);
// a caller that keeps the composite table itself (ljmd_multi.cpp): the ids start again at rank S + j
int migrate_rebase(ljmd_t *h);

}  // namespace ljmdh
#endif
