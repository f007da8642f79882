// ljmd_capi.cpp -- host side of libljmd.so: the C ABI declared in include/ljmd.h.
//
// Owns the HBM-resident simulation state and sequences the gfx950 kernels of
// ljmd_kernels.hip / ljmd_sort.hip on one HIP stream.  There is no CPU compute path in
// this library: without a HIP device every compute entry point returns
// LJMD_ERR_NO_DEVICE.
#include "ljmd_engine.h"
#include "ljmd_multi.h"

namespace ljmdh {

thread_local std::string g_last_error = "";

int fail(const ljmd_t *h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (h) const_cast<ljmd_t *>(h)->err = buf;
    return code;
}

bool fast_path_ok(const ljmd_t *h) { return h->rc_allows_fast && h->positions_compact && !h->force_generic; }

// the buffer of workgroup partials of the current force evaluation (two of them alternate when the record of a step is
// folded by the next step's tail launch)
double *wg_part_now(ljmd_t *h) { return h->d_wg_part + (h->fuse_tail ? (size_t)h->fold_parity * h->wg_part_stride : 0); }

PairArgs pair_args(ljmd_t *h, bool fast)
{
    PairArgs a;
    a.pos = h->d_pos;
    a.slab = h->d_slab;
    a.wg_part = wg_part_now(h);
    a.mask = h->d_mask;
    a.bbox = h->d_bbox;
    a.inline_mask = (fast && h->fuse_tail && !h->use_n3) ? 1 : 0;
    a.rc2_skin = h->rc2 * (1.0 + 1e-10);
    a.n = h->n;
    a.S = h->S;
    a.P = h->P;
    a.G = h->G;
    a.rank = h->rank;
    a.TB = h->TB;
    a.T = h->T;
    a.W = h->W;
    a.chunk = fast ? h->chunk_t : h->chunk_g;
    a.L = h->L;
    a.invL = h->invL;
    a.rc2 = h->rc2;
    return a;
}

GeometryArgs geometry_args(ljmd_t *h)
{
    GeometryArgs a;
    a.pos = h->d_pos;
    a.bbox = h->d_bbox;
    a.pos_tc = h->use_n3 ? h->d_pos_tc : nullptr;
    a.invL = h->invL;
    a.mask = h->d_mask;
    a.P = h->P;
    a.G = h->G;
    a.rank = h->rank;
    a.TB = h->TB;
    a.T = h->T;
    a.W = h->W;
    a.RT = h->rt;
    a.L = h->L;
    a.rc2_skin = h->rc2 * (1.0 + 1e-10);
    a.mask_far = h->d_mask_far;       // NULL unless mixed precision
    a.rsplit2 = h->r_split * h->r_split;
    // 2^(26/3) = 406.3747: r^-6 < 2^-26 beyond it, 406.5 with a margin of 1e-3 for the fp32 roundings of u^3 (pair_n3_f32<., VFAR>);
    // LJMD_FP32_VFAR=0 switches the form off
    a.rvfar2 = env_int("LJMD_FP32_VFAR", 1) != 0 ? 406.5 : HUGE_VAL;
    a.pertile_images = env_int("LJMD_N3_PERTILE", 1) != 0 ? 1 : 0;
    a.both_ties = h->both_ties ? 1 : 0;
    return a;
}

N3Args n3_args(ljmd_t *h)
{
    N3Args a;
    a.pos = h->d_pos_tc;       // every tile in one periodic image (tile_boxes_kernel / the drift kernel's fused form)
    a.mask = h->d_mask;
    a.bbox = h->d_bbox;
    a.desc = h->d_desc;
    a.desc2 = h->d_desc2;
    a.slab_i = h->d_slab;
    a.slab_j = h->d_slab_j;
    a.flag_j = h->d_flag_j;
    a.wg_part = wg_part_now(h);
    a.S = h->S;
    a.P = h->P;
    a.G = h->G;
    a.rank = h->rank;
    a.TB = h->TB;
    a.T = h->T;
    a.W = h->W;
    a.NG = h->NG;
    a.NGo = h->NGo;
    a.Dmax = h->Dmax;
    a.CS = h->CS;
    a.by_group = h->j_by_group;
    a.dchunk = h->dchunk;
    a.uchunk = h->uchunk;
    a.xcd_remap = 0;
    a.inline_class = (h->fuse_tail && h->rt <= 2 && h->wg_waves == 1) ? 1 : 0;
    a.both_ties = h->both_ties ? 1 : 0;
    a.rc2_skin = h->rc2 * (1.0 + 1e-10);
    a.energy = h->want_energy ? 1 : 0;
    a.RT = h->rt;
    a.L = h->L;
    a.invL = h->invL;
    a.rc2 = h->rc2;
    return a;
}

IntegrateArgs integrate_args(ljmd_t *h)
{
    IntegrateArgs a;
    a.r = own_block(h);
    a.ru = h->d_ru;
    a.v = h->d_v;
    a.a = h->d_a;
    a.fsum = needs_force_exchange(h) ? h->d_frecv : h->d_fpart;
    a.bbox = nullptr;
    a.pos_tc = nullptr;
    a.RT = std::max(1, h->rt);
    a.ticket = nullptr;
    a.ke_part = h->d_ke_part;
    a.rows = h->P;
    a.P = h->P;
    a.L = h->L;
    a.invL = h->invL;
    a.dt = h->dt;
    a.dt_half = h->dt_half;
    a.dt_sq_half = h->dt_sq_half;
    return a;
}

ReduceArgs reduce_args(ljmd_t *h, int nslab, bool n3)
{
    ReduceArgs a;
    a.slab = h->d_slab;
    a.slab_j = n3 ? h->d_slab_j : nullptr;
    a.flag_j = n3 ? h->d_flag_j : nullptr;
    a.slab_j2 = (n3 && h->mode == LJMD_PRECISION_FP32_FORCE) ? reinterpret_cast<const float *>(h->d_slab_j2) : nullptr;
    a.flag_j2 = (n3 && h->mode == LJMD_PRECISION_FP32_FORCE) ? h->d_flag_j2 : nullptr;
    a.fpart = h->d_fpart;
    a.nslab = nslab;
    a.P = h->P;
    a.G = h->G;
    a.rank = h->rank;
    a.TB = h->TB;
    a.CS = h->CS;
    a.CS2 = h->CS2;
    a.RT = h->rt;
    return a;
}

FinalizeArgs finalize_args(ljmd_t *h, int n_wg, bool with_ke, double pair_scale)
{
    FinalizeArgs a;
    a.pair_scale = pair_scale;
    a.wg_part = wg_part_now(h);
    a.ke_part = h->d_ke_part;
    a.ke_tile = nullptr;
    a.ring = h->d_ring;
    a.ring_pos = h->d_ring_pos;
    a.n_wg = n_wg;
    a.n_ke = with_ke ? h->n_ke : 0;
    a.ring_cap = kRingCap;
    return a;
}

EventSet *next_events(ljmd_t *h)
{
    if (!h->profiling || h->ev_used >= (size_t)kMaxProfiledLaunches) return nullptr;
    if (h->ev_used == h->ev_pool.size()) {
        EventSet q;
        for (auto &e : q.e)
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
        h->ev_pool.push_back(q);
    }
    EventSet *q = &h->ev_pool[h->ev_used++];
    q->has_pos_x = q->has_force_x = false;
    return q;
}

// Spatial re-ordering of the owned shard: keys -> stable radix sort -> gather r, ru, v (+ a when
// asked) and compose the slot->original permutation.  Performance only (ljmd_sort.hip).
int resort(ljmd_t *h, bool with_accel)
{
    {
    // recursive median split (ljmd_sort.hip): one composite-key radix sort per level, along the axis
    // chosen for that level at set_state (longest remaining extent of the shard)
    LJMD_HIP(h, launch_iota(h->d_idx, h->P, h->stream));
    LJMD_HIP(h, hipMemcpyAsync(h->d_idx2, h->d_idx, (size_t)h->P * sizeof(int), hipMemcpyDeviceToDevice,
                               h->stream));   // slots S..P-1 (padding) keep their identity in both buffers
    int *cur = h->d_idx, *nxt = h->d_idx2;
    for (size_t l = 0; l < h->kd_level_nseg.size(); ++l) {
        const double *axis = own_block(h) + (size_t)h->kd_axis[l] * h->P;
        LJMD_HIP(h, kd_level(h->d_cub, h->cub_bytes, axis, h->L, h->d_kd_keys, h->d_kd_keys2, cur, nxt, h->S,
                             h->kd_level_nseg[l], h->d_kd_offsets + h->kd_level_off[l], h->stream));
        std::swap(cur, nxt);
    }
    if (cur != h->d_idx2)
        LJMD_HIP(h, hipMemcpyAsync(h->d_idx2, cur, (size_t)h->P * sizeof(int), hipMemcpyDeviceToDevice,
                                   h->stream));
    }
    const size_t bytes3 = 3 * (size_t)h->P * sizeof(double);
    double *sets[4] = {own_block(h), h->d_ru, h->d_v, h->d_a};
    for (int k = 0; k < (with_accel ? 4 : 3); ++k) {
        LJMD_HIP(h, launch_gather3(sets[k], h->d_tmp3, h->d_idx2, h->P, h->stream));
        LJMD_HIP(h, hipMemcpyAsync(sets[k], h->d_tmp3, bytes3, hipMemcpyDeviceToDevice, h->stream));
    }
    LJMD_HIP(h, launch_gather_perm(h->d_perm, h->d_perm2, h->d_idx2, h->P, h->stream));
    std::swap(h->d_perm, h->d_perm2);
    h->perm_dirty = true;
    h->steps_since_sort = 0;
    return LJMD_OK;
}

int refresh_perm(ljmd_t *h)
{
    if (!h->perm_dirty) return LJMD_OK;
    LJMD_HIP(h, hipMemcpyAsync(h->h_perm.data(), h->d_perm, (size_t)h->P * sizeof(int), hipMemcpyDeviceToHost,
                               h->stream));
    LJMD_HIP(h, hipStreamSynchronize(h->stream));
    h->perm_dirty = false;
    return LJMD_OK;
}

// All collectives of the communicator go through ONE stream.  With the communication stream in use
// (LJMD_OVERLAP_EXCHANGE=1, default) a collective is fenced against the engine's stream by two events:
// comm_begin = "the engine's work so far is a prerequisite", comm_end = "the engine's later work waits for it".
bool use_comm_stream(const ljmd_t *h) { return h->overlap_exchange && h->comm_stream != nullptr; }

int comm_begin(ljmd_t *h)
{
    LJMD_HIP(h, hipEventRecord(h->ev_pos_ready, h->stream));
    LJMD_HIP(h, hipStreamWaitEvent(h->comm_stream, h->ev_pos_ready, 0));
    return LJMD_OK;
}

int comm_end(ljmd_t *h)
{
    LJMD_HIP(h, hipEventRecord(h->ev_gather_done, h->comm_stream));
    LJMD_HIP(h, hipStreamWaitEvent(h->stream, h->ev_gather_done, 0));
    return LJMD_OK;
}

int allgather_on(ljmd_t *h, hipStream_t s)
{
    // in place: the send block is this rank's slice of the receive buffer
    const ncclResult_t r = ncclAllGather(own_block(h), h->d_pos, 3 * (size_t)h->P, ncclDouble, h->comm, s);
    if (r != ncclSuccess) return fail(h, LJMD_ERR_HIP, "ncclAllGather failed: %s", ncclGetErrorString(r));
    return LJMD_OK;
}

// Phase A: pair kernel on the exchange buffer + deterministic slab reduction into fpart.
int enqueue_pair_forces(ljmd_t *h, EventSet *q)
{
    if (h->inject_failure_at >= 0 && (int)h->ring_issued == h->inject_failure_at) {
        h->inject_failure_at = -1;
        return fail(h, LJMD_ERR_HIP, "injected failure in the force phase (LJMD_INJECT_FAILURE_AT_STEP)");
    }
    const bool fast = fast_path_ok(h);
    if (q) LJMD_HIP(h, hipEventRecord(q->e[1], h->stream));
    int nslab, n_wg;
    bool n3 = false;
    if (fast) {
        GeometryArgs ga = geometry_args(h);
        if (!h->use_n3) ga.mask_far = nullptr;
        if (!h->boxes_valid) LJMD_HIP(h, launch_tile_boxes(ga, h->stream));
        h->boxes_valid = false;                    // good for this evaluation only
        if (h->fuse_tail)
            ;               // small single-rank system: the pair kernel's waves work their pass descriptors / mask words out themselves
        else if (h->use_n3)      // tile-pair test + pass descriptors of the Newton-3 kernels in one launch (mixed mode: NEAR and FAR)
            LJMD_HIP(h, launch_tile_class(ga, h->invL, h->rc2, h->S, h->NGo, h->d_desc,
                                          h->mode == LJMD_PRECISION_FP32_FORCE ? h->d_desc_far : nullptr, h->d_desc2, h->stream));
        else                // the gather kernel reads the bit mask
            LJMD_HIP(h, launch_tile_mask(ga, h->stream));
        if (q) LJMD_HIP(h, hipEventRecord(q->e[2], h->stream));
        if (h->use_n3) {
            const dim3 grid((h->NGo + h->wg_waves - 1) / h->wg_waves, h->nslab_n);     // wg_waves row groups per workgroup
            N3Args na = n3_args(h);
            // (kXcdMinGroups = 256 row groups per rank: with 4096 column tiles and 256 row groups -- rank
            //  0 of 4 at n = 262144 -- the mapping still saves 4.7 % (tools/probe_rank.py), with 128 it is neutral, as it
            //  is for single-rank systems of 16384..65536 particles)
            na.xcd_remap = (h->xcd_remap > 0 && (int)grid.x >= kXcdMinGroups && grid.x % (8 * h->xcd_remap) == 0) ? h->xcd_remap : 0;
            // mixed precision: the two pair kernels write disjoint slabs and partials -- the far pass (the long one) goes to
            // its own stream first and the near pass, mostly descriptor look-ups with a few passes between them, runs beside it
            const bool far_beside = h->mode == LJMD_PRECISION_FP32_FORCE && h->far_stream != nullptr;
            if (far_beside) {
                LJMD_HIP(h, hipEventRecord(h->ev_far_go, h->stream));
                LJMD_HIP(h, hipStreamWaitEvent(h->far_stream, h->ev_far_go, 0));
            } else {
                LJMD_HIP(h, launch_pair_n3(na, grid, h->wg_waves, h->stream));        // all pairs, or the NEAR ones
            }
            nslab = h->nslab_n;
            n_wg = grid.x * grid.y * h->wg_waves;                                      // one partial per wave
            n3 = true;
            if (h->mode == LJMD_PRECISION_FP32_FORCE) {
                const dim3 fgrid(h->NGo, h->nslab_n);                                  // one wave per workgroup
                // far pass in fp32: its own row-side slices, column-side slab and workgroup partials
                N3Args fa = n3_args(h);
                fa.mask = h->d_mask_far;
                fa.slab_i = h->d_slab + (size_t)h->nslab_n * 3 * h->P;
                fa.slab_j = h->d_slab_j2;
                fa.flag_j = h->d_flag_j2;
                fa.desc = h->d_desc_far;
                fa.CS = h->CS2;
                fa.by_group = h->G > 1 ? 1 : 0;          // one wave per workgroup whatever wg_waves is: block index = offset d
                                                         // (CS2 = Dmax + 1) on one rank, the row group on several (CS2 = NGo)
                fa.xcd_remap = (h->xcd_remap > 0 && (int)fgrid.x >= kXcdMinGroups && fgrid.x % (8 * h->xcd_remap) == 0) ? h->xcd_remap : 0;
                fa.wg_part = h->d_wg_part + 2 * (size_t)n_wg;
                LJMD_HIP(h, launch_pair_n3_f32(fa, fgrid, far_beside ? h->far_stream : h->stream));
                if (far_beside) {
                    LJMD_HIP(h, hipEventRecord(h->ev_far_done, h->far_stream));
                    LJMD_HIP(h, launch_pair_n3(na, grid, h->wg_waves, h->stream));    // the NEAR pairs, beside the far pass
                    LJMD_HIP(h, hipStreamWaitEvent(h->stream, h->ev_far_done, 0));
                }
                nslab *= 2;
                n_wg += fgrid.x * fgrid.y;
            }
        } else {
            const dim3 grid(h->TB / kWavesPerBlock, h->nslab_t);
            LJMD_HIP(h, launch_pair_tiles(pair_args(h, true), grid, h->stream));
            nslab = h->nslab_t;
            n_wg = grid.x * grid.y;
        }
    } else {
        if (h->use_n3 && h->G > 1)
            return fail(h, LJMD_ERR_STATE,
                        "multi-rank Newton-3 run needs wrapped positions and rc <= (1-1e-9) L/2 (set LJMD_N3=0)");
        if (q) LJMD_HIP(h, hipEventRecord(q->e[2], h->stream));
        const dim3 grid(h->P / kBlock, h->nslab_g);
        LJMD_HIP(h, launch_pair_rows_generic(pair_args(h, false), grid, h->stream));
        nslab = h->nslab_g;
        n_wg = grid.x * grid.y;
    }
    if (q) LJMD_HIP(h, hipEventRecord(q->e[3], h->stream));
    h->reduce_deferred = fast && h->fuse_tail && h->kick_hint >= 0 && n_wg <= kDirectFoldMax && !needs_force_exchange(h);
    if (h->reduce_deferred) {            // the tail launch of enqueue_kick reduces, kicks and folds the record in one kernel
        h->deferred_nslab = nslab;
        h->deferred_n3 = n3;
    } else {
        LJMD_HIP(h, launch_reduce_forces(reduce_args(h, nslab, n3), needs_force_exchange(h), h->stream));
    }
    h->forces_pending = true;
    h->pending_n_wg = n_wg;
    h->pending_scale = n3 ? 1.0 : 0.5;
    return LJMD_OK;
}

// Phase B: (multi-rank Newton-3) reduce-scatter of the partial accelerations, then x24, optional
// second half-kick, kinetic-energy partials and this step's partial record.
int enqueue_kick(ljmd_t *h, bool kick, EventSet *q)
{
    if (h->reduce_deferred) {
        h->reduce_deferred = false;
        const bool drift = kick && h->next_drift_hint;
        IntegrateArgs ia = integrate_args(h);
        ia.ticket = h->d_ticket;
        if (drift) {                     // the next step's K1 writes the tile boxes and the coherent copy as well
            ia.bbox = h->d_bbox;
            ia.pos_tc = h->use_n3 ? h->d_pos_tc : nullptr;
        }
        FinalizeArgs fa = finalize_args(h, h->pending_n_wg, kick, h->pending_scale);
        fa.ke_tile = h->d_ke_tile + (size_t)h->fold_parity * 3 * h->T;
        // another step of this batch follows: its tail launch folds this step's record beside its own work
        const bool defer = kick && h->next_drift_hint && h->defer_record;
        FinalizeArgs prev{};
        if (h->fold_pending) prev = h->pending_fold;
        if (defer) ia.ticket = nullptr;
        LJMD_HIP(h, launch_tile_tail(reduce_args(h, h->deferred_nslab, h->deferred_n3), ia, fa, prev, kick, drift, h->stream));
        h->fold_pending = defer;
        h->pending_fold = fa;
        h->fold_parity ^= 1;                 // the next force evaluation writes the other pair of buffers
        if (q) LJMD_HIP(h, hipEventRecord(q->e[4], h->stream));
        h->drift_prefused = drift;
        h->ring_issued++;
        h->have_accel = true;
        h->forces_pending = false;
        return LJMD_OK;
    }
    if (needs_force_exchange(h) && !h->external_force_exchange) {
        if (!h->comm) return fail(h, LJMD_ERR_STATE, "multi-rank Newton-3 step: call ljmd_comm_init first");
        const bool cs = use_comm_stream(h);
        if (cs) {
            const int rc_ = comm_begin(h);
            if (rc_ != LJMD_OK) return rc_;
        }
        const hipStream_t xs = cs ? h->comm_stream : h->stream;
        const size_t blk = 3 * (size_t)h->P;
        if (q) LJMD_HIP(h, hipEventRecord(q->e[7], xs));
        if (h->exchange_alltoall) {
            // every rank sends block g of its fpart straight to rank g (one xGMI link per peer on the fully
            // connected mesh) and adds the G blocks it receives in rank order: explicit, reproducible sum order
            ncclResult_t r = ncclGroupStart();
            for (int g = 0; g < h->G && r == ncclSuccess; ++g) {
                r = ncclSend(h->d_fpart + (size_t)g * blk, blk, ncclDouble, g, h->comm, xs);
                if (r == ncclSuccess) r = ncclRecv(h->d_fall + (size_t)g * blk, blk, ncclDouble, g, h->comm, xs);
            }
            const ncclResult_t e = ncclGroupEnd();
            if (r == ncclSuccess) r = e;
            if (r != ncclSuccess) return fail(h, LJMD_ERR_HIP, "force all-to-all failed: %s", ncclGetErrorString(r));
            LJMD_HIP(h, launch_sum_blocks(h->d_fall, h->d_frecv, h->G, (int)blk, xs));
        } else {
            const ncclResult_t r = ncclReduceScatter(h->d_fpart, h->d_frecv, blk, ncclDouble, ncclSum, h->comm, xs);
            if (r != ncclSuccess) return fail(h, LJMD_ERR_HIP, "ncclReduceScatter failed: %s", ncclGetErrorString(r));
        }
        if (q) {
            LJMD_HIP(h, hipEventRecord(q->e[8], xs));
            q->has_force_x = true;
        }
        if (cs) {
            const int rc_ = comm_end(h);
            if (rc_ != LJMD_OK) return rc_;
        }
    }
    if (h->fold_pending) {               // (a record left to "the next tail launch" that is not coming: append it now)
        LJMD_HIP(h, launch_finalize(h->pending_fold, nullptr, h->stream));
        h->fold_pending = false;
    }
    if (h->fuse_small && h->pending_n_wg <= kDirectFoldMax) {
        IntegrateArgs ia = integrate_args(h);
        ia.ticket = h->d_ticket;
        LJMD_HIP(h, launch_kick_finalize(ia, finalize_args(h, h->pending_n_wg, kick, h->pending_scale), kick, h->stream));
    } else {
        LJMD_HIP(h, launch_kick(integrate_args(h), kick, h->stream));
        LJMD_HIP(h, launch_finalize(finalize_args(h, h->pending_n_wg, kick, h->pending_scale), h->d_fold, h->stream));
    }
    if (q) LJMD_HIP(h, hipEventRecord(q->e[4], h->stream));
    h->ring_issued++;
    h->have_accel = true;
    h->forces_pending = false;
    return LJMD_OK;
}

int enqueue_forces(ljmd_t *h, bool kick, EventSet *q, bool next_drift)
{
    h->kick_hint = kick ? 1 : 0;         // both phases from one caller: the tail launch may take everything behind the pair kernel
    h->next_drift_hint = next_drift;
    int rc_ = enqueue_pair_forces(h, q);
    if (rc_ == LJMD_OK) rc_ = enqueue_kick(h, kick, q);
    h->kick_hint = -1;
    h->next_drift_hint = false;
    h->reduce_deferred = false;
    return rc_;
}

// K1, positions: drift + wrap + unwrapped update.  On a re-sort step (and only then) the whole of K1 runs here, followed
// by the re-sort: the velocity half-kick must precede the permutation, so there is nothing left to overlap (*split = false).
int enqueue_drift_positions(ljmd_t *h, EventSet *q, bool *split)
{
    if (q) LJMD_HIP(h, hipEventRecord(q->e[0], h->stream));
    h->gather_done_for_step = false;
    const bool resort_now = h->sort_enabled && fast_path_ok(h) && h->steps_since_sort + 1 >= h->resort_every;
    *split = !resort_now;
    LJMD_HIP(h, launch_drift_kick(integrate_args(h), resort_now ? 0 : 1, h->stream));
    h->positions_compact = true;  // freshly wrapped into [0, L]
    if (h->sort_enabled && fast_path_ok(h) && ++h->steps_since_sort >= h->resort_every) {
        if (*split) {             // (positions that only became compact with this wrap: finish K1 before permuting)
            LJMD_HIP(h, launch_drift_kick(integrate_args(h), 2, h->stream));
            *split = false;
        }
        return resort(h, false);  // a(t) is dead after the drift/kick: K3 rewrites it
    }
    return LJMD_OK;
}

// K1, velocities: the first half-kick (reads a(t), which nothing rewrites before the kick kernel of this step)
int enqueue_drift_velocities(ljmd_t *h)
{
    LJMD_HIP(h, launch_drift_kick(integrate_args(h), 2, h->stream));
    return LJMD_OK;
}

int enqueue_drift(ljmd_t *h, EventSet *q)
{
    const bool collectives = h->comm && (h->G > 1 || h->force_collectives);
    if (collectives && use_comm_stream(h)) {
        // positions first; the all-gather starts on the communication stream as soon as they are final and
        // overlaps the velocity half-kick; the engine's stream resumes (geometry pre-pass, pair kernel) when
        // the gathered positions have arrived.  (Re-sort steps permute the block after K1: the gather then follows
        // serially, ljmd_allgather_positions.)
        bool split = false;
        int rc_ = enqueue_drift_positions(h, q, &split);
        if (rc_ != LJMD_OK || !split) return rc_;
        rc_ = comm_begin(h);
        if (rc_ != LJMD_OK) return rc_;
        if (q) LJMD_HIP(h, hipEventRecord(q->e[5], h->comm_stream));
        rc_ = allgather_on(h, h->comm_stream);
        if (rc_ != LJMD_OK) return rc_;
        if (q) {
            LJMD_HIP(h, hipEventRecord(q->e[6], h->comm_stream));
            q->has_pos_x = true;
        }
        LJMD_HIP(h, hipEventRecord(h->ev_gather_done, h->comm_stream));
        rc_ = enqueue_drift_velocities(h);                                   // runs while the gather is in flight
        if (rc_ != LJMD_OK) return rc_;
        LJMD_HIP(h, hipStreamWaitEvent(h->stream, h->ev_gather_done, 0));
        h->gather_done_for_step = true;
        return LJMD_OK;
    }
    if (q) LJMD_HIP(h, hipEventRecord(q->e[0], h->stream));
    h->gather_done_for_step = false;
    const bool resort_now = h->sort_enabled && fast_path_ok(h) && h->steps_since_sort + 1 >= h->resort_every;
    if (h->drift_prefused) {
        // the previous step's tail launch has already run this K1 (tile_tail_kernel<.., DRIFT>), boxes included
        h->drift_prefused = false;
        h->boxes_valid = !resort_now && fast_path_ok(h);
        h->positions_compact = true;
        if (h->sort_enabled && fast_path_ok(h) && ++h->steps_since_sort >= h->resort_every) return resort(h, false);
        return LJMD_OK;
    }
    IntegrateArgs ia = integrate_args(h);
    // single rank, no re-sort behind this kernel: the drift kernel's waves are the tiles -- let them write
    // the bounding boxes of the new positions and skip tile_boxes_kernel in the force evaluation that follows
    // (only where a launch matters: at n = 262144 the six wave reductions cost the HBM-bound kernel more -- 8.0 ->
    // 11.5 us -- than the 5 us boxes kernel they replace)
    h->boxes_valid = h->fuse_small && h->G == 1 && h->n <= 65536 && !resort_now && fast_path_ok(h);
    if (h->boxes_valid) {
        ia.bbox = h->d_bbox;
        ia.pos_tc = h->use_n3 ? h->d_pos_tc : nullptr;
    }
    LJMD_HIP(h, launch_drift_kick(ia, 0, h->stream));
    h->positions_compact = true;  // freshly wrapped into [0, L]
    if (h->sort_enabled && fast_path_ok(h) && ++h->steps_since_sort >= h->resort_every)
        return resort(h, false);  // a(t) is dead after the drift/kick: K3 rewrites it
    return LJMD_OK;
}

// Reads back the not-yet-consumed partial records (at most kRingCap) into h_ring.
int fetch_ring(ljmd_t *h, unsigned count)
{
    if (count > h->ring_issued - h->ring_consumed)
        return fail(h, LJMD_ERR_STATE, "requested %u step records but only %u are pending", count,
                    h->ring_issued - h->ring_consumed);
    h->ring_consumed = h->ring_issued - count;  // older unread records are dropped
    unsigned done = 0;
    while (done < count) {
        const unsigned pos = (h->ring_consumed + done) % kRingCap;
        const unsigned run = std::min(count - done, kRingCap - pos);
        LJMD_HIP(h, hipMemcpyAsync(h->h_ring + (size_t)done * kPartialStride,
                                   h->d_ring + (size_t)pos * kPartialStride,
                                   (size_t)run * kPartialStride * sizeof(double),
                                   hipMemcpyDeviceToHost, h->stream));
        done += run;
    }
    LJMD_HIP(h, hipStreamSynchronize(h->stream));
    h->ring_consumed = h->ring_issued;
    return LJMD_OK;
}

void combine_one(const ljmd_t *h, const double *recs, int n_ranks, double *epot, double *ekin,
                 double *d_epot, double *dd_epot)
{
    double s12 = 0, s6 = 0, kx = 0, ky = 0, kz = 0;
    for (int g = 0; g < n_ranks; ++g) {  // fixed rank order
        const double *r = recs + (size_t)g * kPartialStride;
        s12 += r[0];
        s6 += r[1];
        kx += r[2];
        ky += r[3];
        kz += r[4];
    }
    // (the kernels already normalised s12, s6 to unordered-pair sums: FinalizeArgs::pair_scale)
    // tail corrections: the reference's compile-time switch use_tail_corrections (lj_potential_energy.f90:36,205-219)
    const double te = h->tail_on ? h->tail_e : 0.0, td = h->tail_on ? h->tail_d : 0.0, tdd = h->tail_on ? h->tail_dd : 0.0;
    if (epot) *epot = 4.0 * (s12 - s6) + te;                          // :140,:188,:221
    if (d_epot) *d_epot = 24.0 * (-2.0 * s12 + s6) + td;              // :143,:177,:192,:222
    if (dd_epot) *dd_epot = 24.0 * (26.0 * s12 - 7.0 * s6) + tdd;     // :178,:193,:223
    if (ekin) *ekin = 0.5 * (kx + ky + kz);                           // verlet.f90:93-95
}

void release(ljmd_t *h)
{
    if (!h) return;
    if (h->device >= 0) (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
    if (h->comm) (void)ncclCommDestroy(h->comm);
    if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
    if (h->far_stream) {
        (void)hipStreamSynchronize(h->far_stream);
        (void)hipStreamDestroy(h->far_stream);
    }
    if (h->ev_far_go) (void)hipEventDestroy(h->ev_far_go);
    if (h->ev_far_done) (void)hipEventDestroy(h->ev_far_done);
    if (h->ev_pos_ready) (void)hipEventDestroy(h->ev_pos_ready);
    if (h->ev_gather_done) (void)hipEventDestroy(h->ev_gather_done);
    for (auto &q : h->ev_pool)
        for (auto &e : q.e) (void)hipEventDestroy(e);
    void *dev[] = {h->d_pos, h->d_ru, h->d_v, h->d_a, h->d_slab, h->d_wg_part, h->d_ke_part, h->d_ring,
                   h->d_ring_pos, h->d_bbox, h->d_mask, h->d_idx, h->d_idx2,
                   h->d_perm, h->d_perm2, h->d_tmp3, h->d_cub, h->d_slab_j, h->d_flag_j, h->d_fpart, h->d_frecv, h->d_fall,
                   h->d_kd_offsets, h->d_kd_keys, h->d_kd_keys2, h->d_mask_far, h->d_slab_j2, h->d_flag_j2, h->d_fold, h->d_ticket,
                   h->d_desc, h->d_desc_far, h->d_desc2, h->d_ke_tile, h->d_pos_tc, h->d_gid0, h->d_mig, h->d_mig_idx, h->d_mig_idx2, h->d_mig_keys,
                   h->d_mig_keys2, h->d_mig_offsets, h->d_mig_cub};
    for (void *p : dev) (void)hipFree(p);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->h_ring) (void)hipHostFree(h->h_ring);
    if (h->copy_stream) {
        (void)hipStreamSynchronize(h->copy_stream);
        (void)hipStreamDestroy(h->copy_stream);
    }
    if (h->ev_snap_ready) (void)hipEventDestroy(h->ev_snap_ready);
    if (h->ev_snap_done) (void)hipEventDestroy(h->ev_snap_done);
    if (h->d_snap) (void)hipFree(h->d_snap);
    if (h->d_snap_perm) (void)hipFree(h->d_snap_perm);
    if (h->h_snap) (void)hipHostFree(h->h_snap);
    if (h->h_snap_perm) (void)hipHostFree(h->h_snap_perm);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// HBM -> host of any of r, ru, v, a (dsts[3 w + k], NULL = skip) and of the last n_records scalar records, all
// behind ONE stream synchronisation; the arrays are delivered in the caller's original particle order.
int download_state(ljmd_t *h, double *const dsts[12], unsigned n_records)
{
    const size_t P = h->P;
    const double *srcs[4] = {own_block(h), h->d_ru, h->d_v, h->d_a};
    bool want[4];
    for (int w = 0; w < 4; ++w) {
        want[w] = dsts[3 * w] || dsts[3 * w + 1] || dsts[3 * w + 2];
        if (want[w])
            LJMD_HIP(h, hipMemcpyAsync(h->h_stage + (size_t)w * 3 * P, srcs[w], 3 * P * sizeof(double),
                                       hipMemcpyDeviceToHost, h->stream));
    }
    if (h->perm_dirty)
        LJMD_HIP(h, hipMemcpyAsync(h->h_perm.data(), h->d_perm, P * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (n_records > 0) {
        const int rc_ = fetch_ring(h, n_records);      // synchronises the stream
        if (rc_ != LJMD_OK) return rc_;
    } else {
        LJMD_HIP(h, hipStreamSynchronize(h->stream));
    }
    h->perm_dirty = false;
    for (int w = 0; w < 4; ++w) {
        if (!want[w]) continue;
        for (int k = 0; k < 3; ++k) {
            double *dst = dsts[3 * w + k];
            if (!dst) continue;
            const double *st = h->h_stage + ((size_t)w * 3 + k) * P;
            for (int i = 0; i < h->P; ++i) {
                const int o = h->h_perm[i];
                if (o < h->S) dst[o] = st[i];           // slot -> original index of the shard
            }
        }
    }
    return LJMD_OK;
}

// stage[ax*P + slot] for the owned shard, slot order = current device order
void stage_permuted(ljmd_t *h, const double *x, const double *y, const double *z, size_t off, double pad)
{
    const double *src[3] = {x + off, y + off, z + off};
    for (int ax = 0; ax < 3; ++ax) {
        double *dst = h->h_stage + (size_t)ax * h->P;
        if (h->migrated) {
            // after an ownership migration the shard is a SET of the caller's particles, not an index range: position o
            // of the engine's order holds particle h_gid0[o] of the arrays given to ljmd_set_state
            const double *glob = src[ax] - off;
            for (int i = 0; i < h->P; ++i) {
                const int o = h->h_perm[i];
                dst[i] = (o < h->S) ? glob[h->h_gid0[o]] : pad;
            }
            continue;
        }
        for (int i = 0; i < h->P; ++i) {
            const int o = h->h_perm[i];
            dst[i] = (o < h->S) ? src[ax][o] : pad;
        }
    }
}

int upload_shard3(ljmd_t *h, double *dst, const double *x, const double *y, const double *z)
{
    int rc_ = refresh_perm(h);
    if (rc_ != LJMD_OK) return rc_;
    stage_permuted(h, x, y, z, (size_t)h->rank * h->S, 0.0);
    LJMD_HIP(h, hipMemcpyAsync(dst, h->h_stage, 3 * (size_t)h->P * sizeof(double), hipMemcpyHostToDevice,
                               h->stream));
    LJMD_HIP(h, hipStreamSynchronize(h->stream));  // staging buffer is reused
    return LJMD_OK;
}

// ---- ownership migration ---------------------------------------------------------------------------------------------
namespace {
int migrate_prepare(ljmd_t *h)
{
    if (h->d_mig) return LJMD_OK;
    const size_t n = (size_t)h->n;
    // levels of the deal: segments = runs of whole shards, halved (lower half = ceil) until every segment is one shard.
    // LJMD_MIGRATE_DEAL=slabs (default): every level splits along x -- G slabs.  With rc ~ L/2 every rank needs every
    // position whatever the shape, so a compact surface buys nothing, while slabs are translation-symmetric in the
    // periodic box: the pair kernel's ownership rule (row group A owns the groups up to half the ring ahead) then gives
    // every rank the same work.  =blocks: the longest remaining extent (2 x 2 x 2 near-cubic blocks at G = 8) -- same total
    // work, but a rank's partners ahead on the ring are face, edge or corner neighbours depending on the rank: measured
    // 2.50 / 2.28 / 2.05 / 1.85 ms per rank (max / mean = 1.16) against 2.14-2.18 for slabs at n = 262144
    // (profiles/r03_deal_shapes_per_rank.txt).
    const char *deal = std::getenv("LJMD_MIGRATE_DEAL");
    const bool blocks = deal && std::strcmp(deal, "blocks") == 0;
    std::vector<int> offsets, bounds = {0, h->G};
    double ext[3] = {h->L, h->L, h->L};
    while (true) {
        bool any = false;
        for (size_t j = 0; j + 1 < bounds.size(); ++j) any = any || (bounds[j + 1] - bounds[j] > 1);
        if (!any) break;
        h->mig_level_off.push_back(offsets.size());
        h->mig_level_nseg.push_back((int)bounds.size() - 1);
        for (int b : bounds) offsets.push_back(b * h->S);
        int best = 0;
        if (blocks)
            for (int ax = 1; ax < 3; ++ax)
                if (ext[ax] > ext[best] * (1.0 + 1e-9)) best = ax;
        h->mig_axis.push_back(best);
        ext[best] *= 0.5;
        std::vector<int> next;
        for (size_t j = 0; j + 1 < bounds.size(); ++j) {
            next.push_back(bounds[j]);
            const int t = bounds[j + 1] - bounds[j];
            if (t > 1) next.push_back(bounds[j] + (t + 1) / 2);
        }
        next.push_back(h->G);
        bounds.swap(next);
    }
    for (int ax = 0; ax < 3; ++ax) h->mig_ext[ax] = ext[ax];
    LJMD_HIP(h, hipMalloc(&h->d_mig_idx, n * sizeof(int)));
    LJMD_HIP(h, hipMalloc(&h->d_mig_idx2, n * sizeof(int)));
    LJMD_HIP(h, hipMalloc(&h->d_mig_keys, n * sizeof(unsigned long long)));
    LJMD_HIP(h, hipMalloc(&h->d_mig_keys2, n * sizeof(unsigned long long)));
    h->mig_cub_bytes = kd_temp_bytes(h->n);
    LJMD_HIP(h, hipMalloc(&h->d_mig_cub, std::max<size_t>(h->mig_cub_bytes, 16)));
    LJMD_HIP(h, hipMalloc(&h->d_mig_offsets, std::max<size_t>(offsets.size(), 2) * sizeof(int)));
    if (!offsets.empty())
        LJMD_HIP(h, hipMemcpyAsync(h->d_mig_offsets, offsets.data(), offsets.size() * sizeof(int), hipMemcpyHostToDevice,
                                   h->stream));
    LJMD_HIP(h, hipStreamSynchronize(h->stream));      // `offsets` goes out of scope
    LJMD_HIP(h, hipMalloc(&h->d_mig, (size_t)h->G * kMigrateRows * h->P * sizeof(double)));   // last: marks "prepared"
    return LJMD_OK;
}
}  // namespace

double *migrate_buffer(ljmd_t *h) { return h->d_mig; }

int migrate_pack(ljmd_t *h)
{
    const int rc_ = migrate_prepare(h);
    if (rc_ != LJMD_OK) return rc_;
    LJMD_HIP(h, launch_migrate_pack(h->d_ru, h->d_v, h->d_a, h->d_perm, h->d_gid0,
                                    h->d_mig + (size_t)h->rank * kMigrateRows * h->P, h->S, h->P, h->stream));
    return LJMD_OK;
}

int migrate_deal(ljmd_t *h)
{
    if (!h->d_mig) return fail(h, LJMD_ERR_STATE, "migrate_deal: migrate_pack has not run");
    LJMD_HIP(h, launch_iota_blocked(h->d_mig_idx, h->n, h->S, h->P, h->stream));
    int *cur = h->d_mig_idx, *nxt = h->d_mig_idx2;
    for (size_t l = 0; l < h->mig_level_nseg.size(); ++l) {
        LJMD_HIP(h, kd_level_blocked(h->d_mig_cub, h->mig_cub_bytes, h->d_pos, h->mig_axis[l], h->P, h->L, h->d_mig_keys,
                                     h->d_mig_keys2, cur, nxt, h->n, h->mig_level_nseg[l],
                                     h->d_mig_offsets + h->mig_level_off[l], h->stream));
        std::swap(cur, nxt);
    }
    LJMD_HIP(h, launch_migrate_select(h->d_pos, h->d_mig, cur + (size_t)h->rank * h->S, h->d_tmp3, h->d_ru, h->d_v, h->d_a,
                                      h->d_gid0, h->S, h->P, h->stream));
    LJMD_HIP(h, hipMemcpyAsync(own_block(h), h->d_tmp3, 3 * (size_t)h->P * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    // the new members in the order of the deal ARE the engine's original order now: identity permutation, then the
    // shard's own k-d order (split axes from the block's extents, as ljmd_set_state chooses them)
    LJMD_HIP(h, launch_iota(h->d_perm, h->P, h->stream));
    for (int i = 0; i < h->P; ++i) h->h_perm[i] = i;
    h->perm_dirty = false;
    {
        double ext[3] = {h->mig_ext[0], h->mig_ext[1], h->mig_ext[2]};
        h->kd_axis.assign(h->kd_level_nseg.size(), 0);
        for (size_t l = 0; l < h->kd_axis.size(); ++l) {
            int best = 0;
            for (int ax = 1; ax < 3; ++ax)
                if (ext[ax] > ext[best] * (1.0 + 1e-9)) best = ax;
            h->kd_axis[l] = best;
            ext[best] *= 0.5;
        }
    }
    h->boxes_valid = false;
    h->h_gid0.resize(h->P);
    LJMD_HIP(h, hipMemcpyAsync(h->h_gid0.data(), h->d_gid0, (size_t)h->P * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (h->sort_enabled && fast_path_ok(h)) {
        const int rc_ = resort(h, true);               // a(t) is live between two steps
        if (rc_ != LJMD_OK) return rc_;
    }
    LJMD_HIP(h, hipStreamSynchronize(h->stream));
    h->migrated = true;
    ++h->migrations;
    return LJMD_OK;
}

int migrate_rebase(ljmd_t *h)
{
    LJMD_HIP(h, launch_iota_offset(h->d_gid0, h->S, h->P, h->rank * h->S, h->stream));
    h->migrated = false;
    return LJMD_OK;
}

}  // namespace ljmdh

// ---------------------------------------------------------------------------
extern "C" {

const char *ljmd_version(void) { return "ljmd 0.6.0 gfx950"; }

int32_t ljmd_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

const char *ljmd_last_error(const ljmd_t *h) { return h ? h->err.c_str() : g_last_error.c_str(); }

int ljmd_create(ljmd_t **out, int32_t n, double box_length, double dt, double rc,
                int32_t precision_mode, int32_t device, int32_t rank, int32_t n_ranks)
{
    if (!out) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_create: out is NULL");
    *out = nullptr;
    // guards of md_types.f90:143-161 and allocate_state :192
    if (n <= 0) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_create: n must be > 0");
    if (!(box_length > 0.0)) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_create: box_length must be > 0");
    if (!(rc > 0.0)) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_create: rc must be > 0");
    if (rc >= 0.5 * box_length)
        return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_create: rc must be < L/2 (minimum image convention)");
    if (!(dt > 0.0)) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_create: dt must be > 0");
    if (precision_mode != LJMD_PRECISION_FP64 && precision_mode != LJMD_PRECISION_FP32_FORCE)
        return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_create: precision_mode %d not available", precision_mode);
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_create: bad rank %d of %d", rank, n_ranks);
    if (n % n_ranks != 0)
        return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_create: n=%d not divisible by n_ranks=%d", n, n_ranks);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, LJMD_ERR_NO_DEVICE, "ljmd_create: no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev)
        return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_create: device %d out of range (0..%d)", device, ndev - 1);

    ljmd_t *h = new (std::nothrow) ljmd;
    if (!h) return fail(nullptr, LJMD_ERR_ALLOC, "ljmd_create: out of host memory");
    h->n = n;
    h->G = n_ranks;
    h->rank = rank;
    h->S = n / n_ranks;
    h->P = ((h->S + kSlotAlign - 1) / kSlotAlign) * kSlotAlign;
    h->TB = h->P / kTile;
    h->T = h->G * h->TB;
    h->W = (h->T + 63) / 64;
    h->device = device;
    h->mode = precision_mode;
    // compute_derived_params, md_types.f90:137-159, same expressions
    h->L = box_length;
    h->invL = 1.0 / box_length;
    h->volume = box_length * box_length * box_length;
    h->rc = rc;
    h->rc2 = rc * rc;
    h->dt = dt;
    h->dt_half = 0.5 * dt;
    h->dt_sq_half = h->dt_half * dt;
    {   // tail corrections, lj_potential_energy.f90:205-223
        const double npd = (double)n;
        const double rc3 = (rc * rc) * rc;
        const double rc6 = ((rc * rc) * (rc * rc)) * (rc * rc);
        const double tf = 8.0 * kPi * (npd * npd) / (h->volume * rc3);
        h->tail_e = tf * ((1.0 / (3.0 * rc6)) - 1.0) / 3.0;
        h->tail_d = 2.0 * tf * (-2.0 / (3.0 * rc6) + 1.0);
        h->tail_dd = 2.0 * tf * (26.0 / (3.0 * rc6) - 7.0);
    }
    h->rc_allows_fast = rc <= (1.0 - 1e-9) * 0.5 * box_length;
    // below ~16 tiles there is nothing for the tile mask to skip: keep the caller's order
    h->sort_enabled = env_int("LJMD_SORT", 1) != 0 && n >= 1024;
    h->force_generic = env_int("LJMD_FORCE_GENERIC", 0) != 0;
    h->force_collectives = env_int("LJMD_FORCE_COLLECTIVES", 0) != 0;
    h->fuse_small = env_int("LJMD_FUSE", 1) != 0;
    // measured at n = 262144: chunks of 4 consecutive row groups per XCD -3 % pair-kernel time (19.8 -> 19.1 ms; 2: -1 %,
    // 8 / 16 / 32: +-0, one contiguous eighth per XCD: +10 %), -1.5 % at n = 131072 and 524288 (profiles/r02_xcd_remap_and_prefetch.txt)
    h->xcd_remap = std::max(0, env_int("LJMD_N3_XCD_REMAP", 4));
    h->inject_failure_at = env_int("LJMD_INJECT_FAILURE_AT_STEP", -1);
    {
        const char *fx = std::getenv("LJMD_FORCE_EXCHANGE");
        h->exchange_alltoall = fx && std::strcmp(fx, "alltoall") == 0;
    }
    // tiles loosen as the particles diffuse while one re-sort costs ~1.2 ms at n = 262144: the larger the system, the
    // sooner a re-sort pays for itself (pair time per rank ~ n^2 / G, sort time ~ n / G: the ratio depends on n only).
    // (Most of the slow-down once measured between two sorts -- +9 % after 9 steps in the liquid -- came from tiles that
    // straddle a box face; the tile-coherent positions of tile_boxes_kernel removed it, and 10 / 15 / 20 / 30 steps now
    // differ by < 3 % at n = 262144.)
    // Small systems (n <= 8192): one re-sort is ~30 launches = 135 us against a 33 us step, and in 200 steps a particle of
    // the liquid moves ~0.5 sigma against tiles of 4.3 sigma: every 200 steps (20: 25 300 steps/s at n = 4096, 100: 29 500,
    // 200: 30 000, 400: 30 500 -- tools/small_n_rate.py, profiles/r03_small_n_two_launch_step.txt).
    // Round 4 (profiles/r04_resort_interval_mid_n.txt): the same holds up to the end of the two-launch regime -- n = 12 288:
    // 8225 steps/s at 20, 8929 at 200; 16 384: 5875 / 6130 -- where nearly every tile pair is inside the cutoff anyway; 50 up
    // to 40 000 (32 768: 2134 / 2158), 20 beyond (65 536: 657 at 20, 645 at 200).
    h->resort_every = std::max(1, env_int("LJMD_RESORT_EVERY", n >= 1000000 ? 5 : n >= 131072 ? 10 : n > 40000 ? 20
                                                                   : n > kFuseTailMaxN ? 50 : 200));
    std::vector<int> kd_offsets;
    {   // k-d levels: segments = runs of whole tiles, halved until every segment is one tile
        const int tiles = (h->S + kTile - 1) / kTile;
        std::vector<int> bounds = {0, tiles};
        while (true) {
            bool any = false;
            for (size_t j = 0; j + 1 < bounds.size(); ++j) any = any || (bounds[j + 1] - bounds[j] > 1);
            if (!any) break;
            h->kd_level_off.push_back(kd_offsets.size());
            h->kd_level_nseg.push_back((int)bounds.size() - 1);
            for (int b : bounds) kd_offsets.push_back(std::min(b * kTile, h->S));
            std::vector<int> next;
            for (size_t j = 0; j + 1 < bounds.size(); ++j) {
                next.push_back(bounds[j]);
                const int t = bounds[j + 1] - bounds[j];
                if (t > 1) next.push_back(bounds[j] + (t + 1) / 2);
            }
            next.push_back(tiles);
            bounds.swap(next);
        }
    }

    // launch geometry: rows x slices >= kTargetWorkgroups
    const int row_blocks = h->P / kBlock;
    {   // generic kernel: slices over the n real particles
        int ns = (kTargetWorkgroups + row_blocks - 1) / row_blocks;
        ns = std::max(1, std::min(ns, (n + 63) / 64));
        h->chunk_g = ((n + ns - 1) / ns + 7) / 8 * 8;
        h->nslab_g = (n + h->chunk_g - 1) / h->chunk_g;
    }
    {   // tile (gather) kernel: slices of column tiles
        int ns = (kTargetWorkgroups + row_blocks - 1) / row_blocks;
        ns = std::max(1, std::min(ns, h->T));
        h->chunk_t = (h->T + ns - 1) / ns;
        h->nslab_t = (h->T + h->chunk_t - 1) / h->chunk_t;
    }
    {   // Newton-3 kernel: NG row groups over all ranks, NGo owned; offsets 0..Dmax in slices
        // tiles per row group: 4 is the measured optimum when there is plenty of work; small systems take 2 or
        // 1 so that (row groups) x (offsets) still fills the 1024 SIMDs
        const bool mixed_mode = precision_mode == LJMD_PRECISION_FP32_FORCE;
        int rt = env_int("LJMD_N3_ROW_TILES", 0);
        if (mixed_mode) rt = kRowTiles;                       // the fp32 far kernel is built for 4
        if (rt != 1 && rt != 2 && rt != kRowTiles) {
            // measured (profiles/r04_unit_sweep.txt; work items cut down to single passes, N3Args::uchunk): 4 wins from
            // n = 32768 up, 2 from 6144 (two-launch step included), 1 below
            auto items = [&](int cand) { const long ngo = h->TB / cand; return ngo * ((long)h->G * ngo / 2 + 1); };
            rt = items(kRowTiles) >= kN3ItemsFor4 ? kRowTiles : items(2) >= kN3ItemsFor2 ? 2 : 1;
        }
        h->rt = rt;
        h->NGo = h->TB / rt;
        h->NG = h->G * h->NGo;
        h->Dmax = h->NG / 2;
        // waves (= consecutive row groups) per pair-kernel workgroup: their column-side partial accelerations are
        // combined in LDS, so the column slab holds one block per (workgroup, column tile) -- wg_waves times less
        // memory and traffic.  Only for 4-tile row groups with plenty of them.
        // Measured at n = 262144 (profiles/r02_wg_waves_lds_combine.txt): the lock step costs more than the smaller
        // slab saves -- pair kernel 18.0 / 18.8 / 19.9 ms, slab reduction 0.61 / 0.39 / 0.28 ms for 1 / 2 / 4 --
        // so the default stays 1 and a larger value is chosen only where the column slab would not fit a budget
        // (LJMD_SLAB_BUDGET_GB, default 64 of the card's 288 GB: n = 1 048 576 on ONE GPU keeps W = 1 with a 52 GB slab --
        // pair + reduction 287 ms against 303 ms with W = 4 and 13 GB -- and 2 097 152 particles run with W = 4, 52 GB).
        int wg = env_int("LJMD_N3_WG_WAVES", 0);
        if (wg != 1 && wg != 2 && wg != 4) {
            const double budget = 1e9 * std::max(1, env_int("LJMD_SLAB_BUDGET_GB", 64));
            const double full = (double)h->T * (h->G > 1 ? h->NGo : h->Dmax + 1) * 3.0 * kTile * sizeof(double);   // slab_j at wg = 1
            wg = full <= budget ? 1 : full <= 2.0 * budget ? 2 : 4;
        }
        if (rt != kRowTiles || h->NGo < 16 * wg) wg = 1;
        h->wg_waves = wg;
        // the tie d = NG / 2 worked from both sides (N3Args::both_ties): equal work for every row group where there are few
        // of them; one rank, one wave per workgroup, fp64 mode
        h->both_ties = wg == 1 && n_ranks == 1 && !mixed_mode && h->NG <= kBothTiesMaxGroups && env_int("LJMD_N3_BOTH_TIES", 1) != 0;
        // slab_j: the blocks of a column tile lie together (N3Args::slab_j)
        h->j_by_group = (h->G > 1 || h->NG % wg != 0) ? 1 : 0;
        h->CS = h->j_by_group ? (h->NGo + wg - 1) / wg : (h->Dmax + wg - 1) / wg + 1;
        h->CS2 = h->G > 1 ? h->NGo : h->Dmax + 1;              // far pass: one wave per workgroup
        const int n3_min = env_int("LJMD_N3_MIN_N", 4096);
        // (rc within 1e-9 of L/2 -- the reference accepts rc_over_L up to 0.5 and rejects only rc >= L/2 -- takes the exact
        //  generic kernel, which has no Newton-3 form: a multi-rank run then needs no force exchange at all, and every
        //  rank must know that when it allocates)
        h->use_n3 = env_int("LJMD_N3", 1) != 0 && n >= n3_min && (h->G == 1 || h->rc_allows_fast);
        // work items = (row group, slice of its units), N3Args::uchunk.  Large systems: slices of whole offsets (dchunk),
        // ~target_waves items; a system with fewer (row group, offset) pairs than that is cut finer, down to one pass per item.
        const int n_off = h->Dmax + h->wg_waves;          // offsets a workgroup walks (relative to its first row group)
        const int n_units = n_off * rt;
        // (a rank of a multi-rank run always aims at 131 072: its kernel is 1 / G of a large system's, measured best with the
        //  most items -- profiles/r02_per_rank_xcd_threshold_and_target_waves.txt, r04_per_rank_work_items.txt)
        const bool plenty = (long)h->NGo * n_off >= kN3LargeItems || n_ranks > 1;
        const int target_waves = std::max(1, env_int("LJMD_N3_TARGET_WAVES", plenty ? 131072 : kN3MidTargetItems));
        const int ns_wanted = (target_waves + h->NGo - 1) / h->NGo;
        int ns = std::max(1, std::min(ns_wanted, n_off));
        // small single-rank systems: at most kDirectFoldMax work items, so that the step record is folded by ONE block
        // whichever way the step is launched (the fused step kernel keeps finalize_body's summation order, not fold_partials')
        const bool small_single = n_ranks == 1 && n <= kFuseTailMaxN && rt <= kFuseTailMaxRowTiles;
        const int ns_cap = small_single ? std::max(1, kDirectFoldMax / std::max(1, h->NGo)) : n_units;
        ns = std::min(ns, ns_cap);
        h->dchunk = (n_off + ns - 1) / ns;
        h->uchunk = h->dchunk * rt;
        if (!mixed_mode && ns_wanted > n_off) {                          // finer than whole offsets
            const int nsu = std::max(1, std::min(std::min(ns_wanted, ns_cap), n_units));
            h->uchunk = (n_units + nsu - 1) / nsu;
        }
        h->nslab_n = (n_units + h->uchunk - 1) / h->uchunk;
    }
    // two launches per step for small single-rank systems (tile_tail_kernel; ljmd_engine.h: fuse_tail)
    h->fuse_tail = h->fuse_small && env_int("LJMD_FUSE_TAIL", 1) != 0 && n_ranks == 1 && n <= kFuseTailMaxN && h->rc_allows_fast &&
                   precision_mode == LJMD_PRECISION_FP64 && (!h->use_n3 || (h->rt <= kFuseTailMaxRowTiles && h->wg_waves == 1));
    h->defer_record = env_int("LJMD_FUSE_DEFER_RECORD", 1) != 0;
    const bool mixed = precision_mode == LJMD_PRECISION_FP32_FORCE;
    if (mixed && (!h->use_n3 || n < kMixedMinN)) {
        // the fp32 far kernel works on 4-tile row groups and only pays where most pairs are far pairs
        delete h;
        return fail(nullptr, LJMD_ERR_INVALID_ARG,
                    "ljmd_create: LJMD_PRECISION_FP32_FORCE needs the Newton-3 path and n >= %d", kMixedMinN);
    }
    {
        const char *rs = std::getenv("LJMD_FP32_SPLIT");
        if (rs && *rs) h->r_split = std::max(0.0, std::atof(rs));
    }
    const int nslab_max = std::max(std::max(h->nslab_g, h->nslab_t), h->use_n3 ? h->nslab_n * (mixed ? 2 : 1) : 1);
    const int n_wg_max = std::max(row_blocks * std::max(h->nslab_g, h->nslab_t),
                                  (h->NGo + 4) * h->nslab_n * (mixed ? 2 : 1));
    h->n_ke = row_blocks;
    h->h_perm.resize(h->P);
    for (int i = 0; i < h->P; ++i) h->h_perm[i] = i;

    auto body = [&]() -> int {
        LJMD_HIP(h, hipSetDevice(device));
        LJMD_HIP(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        const size_t P3 = 3 * (size_t)h->P * sizeof(double);
        LJMD_HIP(h, hipMalloc(&h->d_pos, P3 * h->G));
        LJMD_HIP(h, hipMalloc(&h->d_ru, P3));
        LJMD_HIP(h, hipMalloc(&h->d_v, P3));
        LJMD_HIP(h, hipMalloc(&h->d_a, P3));
        LJMD_HIP(h, hipMalloc(&h->d_tmp3, P3));
        LJMD_HIP(h, hipMalloc(&h->d_slab, P3 * nslab_max));
        h->wg_part_stride = 2 * (size_t)n_wg_max;
        LJMD_HIP(h, hipMalloc(&h->d_wg_part, (h->fuse_tail ? 2 : 1) * h->wg_part_stride * sizeof(double)));
        if (h->use_n3) {
            const size_t n_blk = (size_t)h->T * h->CS;
            LJMD_HIP(h, hipMalloc(&h->d_slab_j, n_blk * 3 * kTile * sizeof(double)));
            LJMD_HIP(h, hipMalloc(&h->d_flag_j, n_blk));
            LJMD_HIP(h, hipMemsetAsync(h->d_flag_j, 0, n_blk, h->stream));
            LJMD_HIP(h, hipMalloc(&h->d_desc, (size_t)h->NGo * h->T * sizeof(unsigned)));
            // cluster passes (ljmd_kernels.hip: n3_cluster_pass): 4-tile row groups, one wave per workgroup
            if (h->rt == kRowTiles && h->wg_waves == 1 && env_int("LJMD_N3_CLUSTERS", 1) != 0)
                LJMD_HIP(h, hipMalloc(&h->d_desc2, (size_t)h->NGo * h->T * 8 * sizeof(float)));
            LJMD_HIP(h, hipMalloc(&h->d_pos_tc, P3 * h->G));
        }
        if (mixed && env_int("LJMD_FP32_FAR_STREAM", 1) != 0) {
            LJMD_HIP(h, hipStreamCreateWithFlags(&h->far_stream, hipStreamNonBlocking));
            LJMD_HIP(h, hipEventCreateWithFlags(&h->ev_far_go, hipEventDisableTiming));
            LJMD_HIP(h, hipEventCreateWithFlags(&h->ev_far_done, hipEventDisableTiming));
        }
        if (mixed) {
            LJMD_HIP(h, hipMalloc(&h->d_mask_far, (size_t)h->TB * h->W * sizeof(uint64_t)));
            LJMD_HIP(h, hipMalloc(&h->d_desc_far, (size_t)h->NGo * h->T * sizeof(unsigned)));
            const size_t n_blk2 = (size_t)h->T * h->CS2;
            LJMD_HIP(h, hipMalloc(&h->d_slab_j2, n_blk2 * 3 * kTile * sizeof(float)));     // fp32 blocks (pair_n3_f32_kernel)
            LJMD_HIP(h, hipMalloc(&h->d_flag_j2, n_blk2));
            LJMD_HIP(h, hipMemsetAsync(h->d_flag_j2, 0, n_blk2, h->stream));
        }
        LJMD_HIP(h, hipMalloc(&h->d_fpart, P3 * (needs_force_exchange(h) ? h->G : 1)));
        if (needs_force_exchange(h)) LJMD_HIP(h, hipMalloc(&h->d_frecv, P3));
        if (needs_force_exchange(h) && h->exchange_alltoall) LJMD_HIP(h, hipMalloc(&h->d_fall, P3 * h->G));
        LJMD_HIP(h, hipMalloc(&h->d_ke_part, 3 * (size_t)h->n_ke * sizeof(double)));
        LJMD_HIP(h, hipMalloc(&h->d_fold, 2 * (size_t)kFoldBlocks * sizeof(double)));
        LJMD_HIP(h, hipMalloc(&h->d_ticket, sizeof(unsigned)));
        LJMD_HIP(h, hipMemsetAsync(h->d_ticket, 0, sizeof(unsigned), h->stream));
        if (h->fuse_tail) {
            LJMD_HIP(h, hipMalloc(&h->d_ke_tile, 2 * 3 * (size_t)h->T * sizeof(double)));        // two buffers, as wg_part
            LJMD_HIP(h, hipMemsetAsync(h->d_ke_tile, 0, 2 * 3 * (size_t)h->T * sizeof(double), h->stream));
        }
        LJMD_HIP(h, hipMalloc(&h->d_ring, (size_t)kRingCap * kPartialStride * sizeof(double)));
        LJMD_HIP(h, hipMalloc(&h->d_ring_pos, sizeof(unsigned)));
        LJMD_HIP(h, hipMalloc(&h->d_bbox, (size_t)h->T * kBoxStride * sizeof(double)));
        LJMD_HIP(h, hipMalloc(&h->d_mask, (size_t)h->TB * h->W * sizeof(uint64_t)));
        LJMD_HIP(h, hipMalloc(&h->d_idx, (size_t)h->P * sizeof(int)));
        LJMD_HIP(h, hipMalloc(&h->d_idx2, (size_t)h->P * sizeof(int)));
        LJMD_HIP(h, hipMalloc(&h->d_perm, (size_t)h->P * sizeof(int)));
        LJMD_HIP(h, hipMalloc(&h->d_perm2, (size_t)h->P * sizeof(int)));
        LJMD_HIP(h, hipMalloc(&h->d_gid0, (size_t)h->P * sizeof(int)));
        LJMD_HIP(h, launch_iota_offset(h->d_gid0, h->S, h->P, h->rank * h->S, h->stream));
        h->cub_bytes = kd_temp_bytes(h->S);
        LJMD_HIP(h, hipMalloc(&h->d_cub, std::max<size_t>(h->cub_bytes, 16)));
        LJMD_HIP(h, hipMalloc(&h->d_kd_keys, (size_t)h->P * sizeof(unsigned long long)));
        LJMD_HIP(h, hipMalloc(&h->d_kd_keys2, (size_t)h->P * sizeof(unsigned long long)));
        LJMD_HIP(h, hipMalloc(&h->d_kd_offsets, std::max<size_t>(kd_offsets.size(), 2) * sizeof(int)));
        if (!kd_offsets.empty())
            LJMD_HIP(h, hipMemcpyAsync(h->d_kd_offsets, kd_offsets.data(), kd_offsets.size() * sizeof(int),
                                       hipMemcpyHostToDevice, h->stream));
        LJMD_HIP(h, hipMemsetAsync(h->d_ring_pos, 0, sizeof(unsigned), h->stream));
        LJMD_HIP(h, hipMemsetAsync(h->d_a, 0, P3, h->stream));
        LJMD_HIP(h, hipMemsetAsync(h->d_ke_part, 0, 3 * (size_t)h->n_ke * sizeof(double), h->stream));
        LJMD_HIP(h, hipHostMalloc(&h->h_stage, P3 * std::max(h->G, 4), hipHostMallocDefault));   // G position blocks, or r, ru, v, a
        LJMD_HIP(h, hipHostMalloc(&h->h_ring, (size_t)kRingCap * kPartialStride * sizeof(double),
                                  hipHostMallocDefault));
        LJMD_HIP(h, hipStreamSynchronize(h->stream));
        return LJMD_OK;
    };
    const int rc_ = body();
    if (rc_ != LJMD_OK) {
        g_last_error = h->err;
        release(h);
        return rc_;
    }
    *out = h;
    return LJMD_OK;
}

void ljmd_destroy(ljmd_t *h)
{
    if (h && h->multi)
        ljmdm::destroy(h);
    else
        release(h);
}

int ljmd_create_multi(ljmd_t **out, int32_t n, double box_length, double dt, double rc, int32_t precision_mode,
                      int32_t n_gpus, const int32_t *devices)
{
    return ljmdm::create(out, n, box_length, dt, rc, precision_mode, n_gpus, devices);
}

// ---- state transfer --------------------------------------------------------

int ljmd_set_state(ljmd_t *h, const double *rx, const double *ry, const double *rz,
                   const double *vx, const double *vy, const double *vz)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_set_state: NULL handle");
    if (!rx || !ry || !rz || !vx || !vy || !vz)
        return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_set_state: NULL array");
    if (h->multi) return ljmdm::set_state(h, rx, ry, rz, vx, vy, vz);
    h->boxes_valid = false;
    h->drift_prefused = false;
    LJMD_HIP(h, hipSetDevice(h->device));
    if (h->poisoned) {
        // a batch of steps failed half-way: drain the stream, forget whatever records were in flight and take the
        // device's own count of finished steps as the truth
        LJMD_HIP(h, hipStreamSynchronize(h->stream));
        LJMD_HIP(h, hipMemcpy(&h->ring_issued, h->d_ring_pos, sizeof(unsigned), hipMemcpyDeviceToHost));
        h->ring_consumed = h->ring_issued;
        h->forces_pending = false;
        h->gather_done_for_step = false;
        h->fold_pending = false;            // (a record a failed batch left to its next tail launch)
        h->poisoned = false;
    }
    const size_t S = h->S, P = h->P;
    // all n positions into the exchange buffer in original order, NaN on the padding;
    // track the coordinate spread (fast-path precondition (a), ljmd_kernels.hip)
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const double *src[3] = {rx, ry, rz};
    bool finite = true;
    for (int g = 0; g < h->G; ++g)
        for (int ax = 0; ax < 3; ++ax) {
            double *dst = h->h_stage + ((size_t)g * 3 + ax) * P;
            const double *s = src[ax] + (size_t)g * S;
            for (size_t i = 0; i < S; ++i) {
                const double x = s[i];
                dst[i] = x;
                lo[ax] = std::min(lo[ax], x);
                hi[ax] = std::max(hi[ax], x);
                finite = finite && std::isfinite(x);
            }
            for (size_t i = S; i < P; ++i) dst[i] = NAN;
        }
    h->positions_compact = finite;
    for (int ax = 0; ax < 3; ++ax)
        if (!(hi[ax] - lo[ax] < 2.4 * h->L)) h->positions_compact = false;
    {   // k-d split axes: halve the longest remaining extent of the OWN shard at every level, so that a
        // slab- or column-shaped shard (multi-GPU index ranges) still ends in near-cubic tiles
        double ext[3];
        for (int ax = 0; ax < 3; ++ax) {
            double slo = INFINITY, shi = -INFINITY;
            const double *sp = src[ax] + (size_t)h->rank * S;
            for (size_t i = 0; i < S; ++i) {
                slo = std::min(slo, sp[i]);
                shi = std::max(shi, sp[i]);
            }
            ext[ax] = std::isfinite(shi - slo) ? std::min(shi - slo, h->L) : h->L;
            if (!(ext[ax] > 0.0)) ext[ax] = 1e-300;
        }
        h->kd_axis.assign(h->kd_level_nseg.size(), 0);
        for (size_t l = 0; l < h->kd_axis.size(); ++l) {
            int best = 0;
            for (int ax = 1; ax < 3; ++ax)
                if (ext[ax] > ext[best] * (1.0 + 1e-9)) best = ax;   // ties -> lowest axis: x, y, z cycling for a cube
            h->kd_axis[l] = best;
            ext[best] *= 0.5;
        }
    }
    LJMD_HIP(h, hipMemcpyAsync(h->d_pos, h->h_stage, 3 * P * h->G * sizeof(double), hipMemcpyHostToDevice,
                               h->stream));
    // ru <- r (md_simulation_program.f90:229-231), own shard
    LJMD_HIP(h, hipMemcpyAsync(h->d_ru, own_block(h), 3 * P * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    LJMD_HIP(h, hipMemsetAsync(h->d_a, 0, 3 * P * sizeof(double), h->stream));
    LJMD_HIP(h, hipStreamSynchronize(h->stream));
    // slot order is the original order again, and the shard is the caller's index range again
    h->migrated = false;
    LJMD_HIP(h, launch_iota_offset(h->d_gid0, h->S, h->P, h->rank * h->S, h->stream));
    for (int i = 0; i < h->P; ++i) h->h_perm[i] = i;
    LJMD_HIP(h, hipMemcpyAsync(h->d_perm, h->h_perm.data(), P * sizeof(int), hipMemcpyHostToDevice, h->stream));
    h->perm_dirty = false;
    int rc_ = upload_shard3(h, h->d_v, vx, vy, vz);
    if (rc_ != LJMD_OK) return rc_;
    h->have_state = true;
    h->have_accel = false;
    if (h->sort_enabled && fast_path_ok(h)) {
        rc_ = resort(h, false);   // accelerations are all zero at this point
        if (rc_ != LJMD_OK) return rc_;
    }
    return LJMD_OK;
}

int ljmd_set_accel(ljmd_t *h, const double *ax, const double *ay, const double *az)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_set_accel: NULL handle");
    if (!ax || !ay || !az) return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_set_accel: NULL array");
    if (h->multi) return ljmdm::set_accel(h, ax, ay, az);
    if (!h->have_state) return fail(h, LJMD_ERR_STATE, "ljmd_set_accel: call ljmd_set_state first");
    LJMD_HIP(h, hipSetDevice(h->device));
    const int rc_ = upload_shard3(h, h->d_a, ax, ay, az);
    if (rc_ == LJMD_OK) h->have_accel = true;
    return rc_;
}

int ljmd_set_unwrapped(ljmd_t *h, const double *ux, const double *uy, const double *uz)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_set_unwrapped: NULL handle");
    if (!ux || !uy || !uz) return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_set_unwrapped: NULL array");
    if (h->multi) return ljmdm::set_unwrapped(h, ux, uy, uz);
    if (!h->have_state) return fail(h, LJMD_ERR_STATE, "ljmd_set_unwrapped: call ljmd_set_state first");
    LJMD_HIP(h, hipSetDevice(h->device));
    return upload_shard3(h, h->d_ru, ux, uy, uz);
}

int ljmd_get_state(ljmd_t *h, double *rx, double *ry, double *rz, double *ux, double *uy, double *uz,
                   double *vx, double *vy, double *vz, double *ax, double *ay, double *az)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_get_state: NULL handle");
    if (!h->have_state) return fail(h, LJMD_ERR_STATE, "ljmd_get_state: no state has been set");
    if (h->multi) {
        double *const p[12] = {rx, ry, rz, ux, uy, uz, vx, vy, vz, ax, ay, az};
        return ljmdm::get_state(h, p);
    }
    LJMD_HIP(h, hipSetDevice(h->device));
    double *const dsts[12] = {rx, ry, rz, ux, uy, uz, vx, vy, vz, ax, ay, az};
    return download_state(h, dsts, 0);
}

// ---- hot path ----------------------------------------------------------------

int ljmd_compute_forces(ljmd_t *h, double *epot, double *d_epot, double *dd_epot)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_compute_forces: NULL handle");
    if (!h->have_state) return fail(h, LJMD_ERR_STATE, "ljmd_compute_forces: no state has been set");
    if (h->poisoned) return fail(h, LJMD_ERR_STATE, "ljmd_compute_forces: handle poisoned by an earlier failure; call ljmd_set_state");
    if (h->multi) return ljmdm::compute_forces(h, epot, d_epot, dd_epot);
    if (h->G != 1)
        return fail(h, LJMD_ERR_STATE, "ljmd_compute_forces: sharded engine; use ljmd_forces_partial");
    LJMD_HIP(h, hipSetDevice(h->device));
    EventSet *q = next_events(h);
    if (q) LJMD_HIP(h, hipEventRecord(q->e[0], h->stream));
    const bool keep = h->want_energy;
    h->want_energy = true;                      // this call exists to return the three sums
    int rc_ = enqueue_forces(h, false, q);
    h->want_energy = keep;
    if (rc_ != LJMD_OK) return rc_;
    rc_ = fetch_ring(h, 1);
    if (rc_ != LJMD_OK) return rc_;
    combine_one(h, h->h_ring, 1, epot, nullptr, d_epot, dd_epot);
    return LJMD_OK;
}

int ljmd_verlet_steps(ljmd_t *h, int32_t nsteps, double *epot, double *ekin, double *d_epot,
                      double *dd_epot)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_verlet_steps: NULL handle");
    if (nsteps < 0) return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_verlet_steps: nsteps < 0");
    if (!h->have_state) return fail(h, LJMD_ERR_STATE, "ljmd_verlet_steps: no state has been set");
    if (!h->have_accel)
        return fail(h, LJMD_ERR_STATE,
                    "ljmd_verlet_steps: accelerations not initialised (call ljmd_compute_forces first)");
    if (h->poisoned) return fail(h, LJMD_ERR_STATE, "ljmd_verlet_steps: handle poisoned by an earlier failure; call ljmd_set_state");
    // nobody reads the potential-energy sums (the warm-up of the initial-configuration driver): forces-only pair kernel
    const bool keep = h->want_energy, wanted = keep && (epot || d_epot || dd_epot);
    if (h->multi) {
        int done_m = 0, rc_ = LJMD_OK;
        ljmdm::set_observables(h, wanted);
        while (done_m < nsteps && rc_ == LJMD_OK) {
            const int batch = std::min<int>(nsteps - done_m, (int)kRingCap);
            rc_ = ljmdm::enqueue_steps(h, batch, false);
            if (rc_ == LJMD_OK)
                rc_ = ljmdm::collect_steps(h, batch, epot ? epot + done_m : nullptr, ekin ? ekin + done_m : nullptr,
                                           d_epot ? d_epot + done_m : nullptr, dd_epot ? dd_epot + done_m : nullptr);
            done_m += batch;
        }
        ljmdm::set_observables(h, keep);
        return rc_;
    }
    if (h->G != 1)
        return fail(h, LJMD_ERR_STATE, "ljmd_verlet_steps: sharded engine; use ljmd_step_begin/finish");
    LJMD_HIP(h, hipSetDevice(h->device));
    int done = 0;
    while (done < nsteps) {
        const int batch = std::min<int>(nsteps - done, (int)kRingCap);
        h->want_energy = wanted;
        for (int s = 0; s < batch; ++s) {
            EventSet *q = next_events(h);
            int rc_ = enqueue_drift(h, q);
            if (rc_ == LJMD_OK) rc_ = enqueue_forces(h, true, q, s + 1 < batch);
            if (rc_ != LJMD_OK) {
                h->want_energy = keep;
                h->poisoned = true;      // a step is half enqueued: no rollback, the state is no longer a trajectory point
                return rc_;
            }
        }
        h->want_energy = keep;
        int rc_ = fetch_ring(h, (unsigned)batch);
        if (rc_ != LJMD_OK) return rc_;
        for (int s = 0; s < batch; ++s)
            combine_one(h, h->h_ring + (size_t)s * kPartialStride, 1, epot ? epot + done + s : nullptr,
                        ekin ? ekin + done + s : nullptr, d_epot ? d_epot + done + s : nullptr,
                        dd_epot ? dd_epot + done + s : nullptr);
        done += batch;
    }
    return LJMD_OK;
}

// ---- asynchronous production loop ---------------------------------------------

namespace {
int enqueue_steps_impl(ljmd_t *h, int32_t nsteps, bool sampled);
}

int ljmd_enqueue_steps(ljmd_t *h, int32_t nsteps)
{
    return enqueue_steps_impl(h, nsteps, false);
}

int ljmd_enqueue_steps_sampled(ljmd_t *h, int32_t nsteps)
{
    return enqueue_steps_impl(h, nsteps, true);
}

int ljmd_set_observables(ljmd_t *h, int32_t on)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_set_observables: NULL handle");
    if (h->multi) return ljmdm::set_observables(h, on != 0);
    h->want_energy = on != 0;
    return LJMD_OK;
}

namespace {
// sampled: only the LAST of the nsteps evaluates the potential-energy sums (the step the reference samples,
// md_simulation_program.f90:361); positions, velocities, accelerations and ekin do not depend on them
int enqueue_steps_impl(ljmd_t *h, int32_t nsteps, bool sampled)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_enqueue_steps: NULL handle");
    if (nsteps < 0) return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_enqueue_steps: nsteps < 0");
    if (!h->have_state) return fail(h, LJMD_ERR_STATE, "ljmd_enqueue_steps: no state has been set");
    if (!h->have_accel)
        return fail(h, LJMD_ERR_STATE,
                    "ljmd_enqueue_steps: accelerations not initialised (call ljmd_compute_forces first)");
    if (h->poisoned) return fail(h, LJMD_ERR_STATE, "ljmd_enqueue_steps: handle poisoned by an earlier failure; call ljmd_set_state");
    if (h->multi) return ljmdm::enqueue_steps(h, nsteps, sampled);
    if (h->G != 1)
        return fail(h, LJMD_ERR_STATE, "ljmd_enqueue_steps: sharded engine; use ljmd_step_begin/finish");
    static_assert(LJMD_MAX_PENDING_STEPS == kRingCap, "LJMD_MAX_PENDING_STEPS out of sync with the record ring");
    if ((h->ring_issued - h->ring_consumed) + (unsigned)nsteps > kRingCap)
        return fail(h, LJMD_ERR_STATE, "ljmd_enqueue_steps: %u + %d pending steps exceed LJMD_MAX_PENDING_STEPS",
                    h->ring_issued - h->ring_consumed, nsteps);
    LJMD_HIP(h, hipSetDevice(h->device));
    const bool keep = h->want_energy;
    for (int s = 0; s < nsteps; ++s) {
        EventSet *q = next_events(h);
        if (sampled) h->want_energy = s == nsteps - 1;
        int rc_ = enqueue_drift(h, q);
        if (rc_ == LJMD_OK) rc_ = enqueue_forces(h, true, q, s + 1 < nsteps);
        if (rc_ != LJMD_OK) {
            h->want_energy = keep;
            h->poisoned = true;
            return rc_;
        }
    }
    h->want_energy = keep;
    return LJMD_OK;
}
}  // namespace

int ljmd_collect_steps(ljmd_t *h, int32_t nsteps, double *epot, double *ekin, double *d_epot, double *dd_epot)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_collect_steps: NULL handle");
    if (nsteps < 0 || nsteps > (int)kRingCap)
        return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_collect_steps: nsteps out of range");
    if (h->poisoned) return fail(h, LJMD_ERR_STATE, "ljmd_collect_steps: handle poisoned by an earlier failure; call ljmd_set_state");
    if (h->multi) return ljmdm::collect_steps(h, nsteps, epot, ekin, d_epot, dd_epot);
    if (h->G != 1)
        return fail(h, LJMD_ERR_STATE, "ljmd_collect_steps: sharded engine; use ljmd_read_partials");
    LJMD_HIP(h, hipSetDevice(h->device));
    const int rc_ = fetch_ring(h, (unsigned)nsteps);
    if (rc_ != LJMD_OK) return rc_;
    for (int s = 0; s < nsteps; ++s)
        combine_one(h, h->h_ring + (size_t)s * kPartialStride, 1, epot ? epot + s : nullptr,
                    ekin ? ekin + s : nullptr, d_epot ? d_epot + s : nullptr, dd_epot ? dd_epot + s : nullptr);
    return LJMD_OK;
}

int ljmd_snapshot_begin(ljmd_t *h)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_snapshot_begin: NULL handle");
    if (!h->have_state) return fail(h, LJMD_ERR_STATE, "ljmd_snapshot_begin: no state has been set");
    if (h->multi) return ljmdm::snapshot_begin(h);
    if (h->snap_in_flight)
        return fail(h, LJMD_ERR_STATE, "ljmd_snapshot_begin: a snapshot is already in flight (call ljmd_snapshot_end)");
    LJMD_HIP(h, hipSetDevice(h->device));
    const size_t P3 = 3 * (size_t)h->P * sizeof(double), PI = (size_t)h->P * sizeof(int);
    if (!h->snap_ready) {
        // lazily, once; a failure part-way leaves snap_ready false and the next call resumes where this one stopped
        // (release() frees whatever exists)
        if (!h->copy_stream) LJMD_HIP(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        if (!h->ev_snap_ready) LJMD_HIP(h, hipEventCreateWithFlags(&h->ev_snap_ready, hipEventDisableTiming));
        if (!h->ev_snap_done) LJMD_HIP(h, hipEventCreateWithFlags(&h->ev_snap_done, hipEventDisableTiming));
        if (!h->d_snap) LJMD_HIP(h, hipMalloc(&h->d_snap, 4 * P3));
        if (!h->d_snap_perm) LJMD_HIP(h, hipMalloc(&h->d_snap_perm, PI));
        if (!h->h_snap) LJMD_HIP(h, hipHostMalloc(&h->h_snap, 4 * P3, hipHostMallocDefault));
        if (!h->h_snap_perm) LJMD_HIP(h, hipHostMalloc(&h->h_snap_perm, PI, hipHostMallocDefault));
        h->snap_ready = true;
    }
    // 1. engine stream: freeze the state as of the steps enqueued so far (HBM -> HBM, ~100 N bytes)
    const double *srcs[4] = {own_block(h), h->d_ru, h->d_v, h->d_a};
    for (int w = 0; w < 4; ++w)
        LJMD_HIP(h, hipMemcpyAsync(h->d_snap + (size_t)w * 3 * h->P, srcs[w], P3, hipMemcpyDeviceToDevice, h->stream));
    LJMD_HIP(h, hipMemcpyAsync(h->d_snap_perm, h->d_perm, PI, hipMemcpyDeviceToDevice, h->stream));
    LJMD_HIP(h, hipEventRecord(h->ev_snap_ready, h->stream));
    // 2. copy stream: HBM -> pinned host, concurrent with whatever the engine stream runs next
    LJMD_HIP(h, hipStreamWaitEvent(h->copy_stream, h->ev_snap_ready, 0));
    LJMD_HIP(h, hipMemcpyAsync(h->h_snap, h->d_snap, 4 * P3, hipMemcpyDeviceToHost, h->copy_stream));
    LJMD_HIP(h, hipMemcpyAsync(h->h_snap_perm, h->d_snap_perm, PI, hipMemcpyDeviceToHost, h->copy_stream));
    LJMD_HIP(h, hipEventRecord(h->ev_snap_done, h->copy_stream));
    h->snap_in_flight = true;
    return LJMD_OK;
}

int ljmd_snapshot_end(ljmd_t *h, double *rx, double *ry, double *rz, double *ux, double *uy, double *uz,
                      double *vx, double *vy, double *vz, double *ax, double *ay, double *az)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_snapshot_end: NULL handle");
    if (h->multi) {
        double *const p[12] = {rx, ry, rz, ux, uy, uz, vx, vy, vz, ax, ay, az};
        return ljmdm::snapshot_end(h, p);
    }
    if (!h->snap_in_flight) return fail(h, LJMD_ERR_STATE, "ljmd_snapshot_end: no snapshot in flight");
    LJMD_HIP(h, hipSetDevice(h->device));
    LJMD_HIP(h, hipEventSynchronize(h->ev_snap_done));      // the transfer only, not the engine's stream
    h->snap_in_flight = false;
    const size_t P = h->P;
    double *dsts[4][3] = {{rx, ry, rz}, {ux, uy, uz}, {vx, vy, vz}, {ax, ay, az}};
    for (int w = 0; w < 4; ++w)
        for (int k = 0; k < 3; ++k) {
            double *dst = dsts[w][k];
            if (!dst) continue;
            const double *st = h->h_snap + ((size_t)w * 3 + k) * P;
            for (size_t i = 0; i < P; ++i) {
                const int o = h->h_snap_perm[i];
                if (o < h->S) dst[o] = st[i];              // slot -> original index of the shard
            }
        }
    return LJMD_OK;
}

int ljmd_kinetic_energy(ljmd_t *h, double *ekin)
{
    if (!h || !ekin) return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_kinetic_energy: NULL argument");
    if (!h->have_state) return fail(h, LJMD_ERR_STATE, "ljmd_kinetic_energy: no state has been set");
    if (h->multi) return ljmdm::kinetic_energy(h, ekin);
    LJMD_HIP(h, hipSetDevice(h->device));
    LJMD_HIP(h, launch_kinetic_fused(integrate_args(h), h->stream));
    std::vector<double> part(3 * (size_t)h->n_ke);
    LJMD_HIP(h, hipMemcpyAsync(part.data(), h->d_ke_part, part.size() * sizeof(double),
                               hipMemcpyDeviceToHost, h->stream));
    LJMD_HIP(h, hipStreamSynchronize(h->stream));
    double s = 0.0;
    for (int b = 0; b < h->n_ke; ++b) s += part[3 * (size_t)b];
    *ekin = 0.5 * s;  // per-rank partial when sharded
    return LJMD_OK;
}

// ---- multi-GPU split phase ---------------------------------------------------

int ljmd_shard_range(const ljmd_t *h, int32_t *i0, int32_t *i1)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_shard_range: NULL handle");
    if (h->multi) return fail(h, LJMD_ERR_STATE, "ljmd_shard_range: a multi-device handle runs the exchange phases itself");
    // after an ownership migration the rank owns a SET of particles, not an index range: stitching rank arrays together
    // by [i0, i1) would silently permute the state
    if (h->migrated)
        return fail(h, LJMD_ERR_STATE, "ljmd_shard_range: this rank has migrated since ljmd_set_state and owns the "
                                       "particles ljmd_particle_ids names, not an index range");
    if (i0) *i0 = h->rank * h->S;
    if (i1) *i1 = (h->rank + 1) * h->S;
    return LJMD_OK;
}

void *ljmd_exchange_buffer(ljmd_t *h, int64_t *n_total, int64_t *own_off, int64_t *own_cnt)
{
    if (!h || h->multi) return nullptr;
    if (n_total) *n_total = 3 * (int64_t)h->P * h->G;
    if (own_off) *own_off = (int64_t)h->rank * 3 * h->P;
    if (own_cnt) *own_cnt = 3 * (int64_t)h->P;
    return h->d_pos;
}

void *ljmd_device_ptr(ljmd_t *h, int32_t which, int32_t axis)
{
    if (!h || h->multi || axis < 0 || axis > 2) return nullptr;
    double *base = nullptr;
    switch (which) {
        case LJMD_R: base = own_block(h); break;
        case LJMD_RU: base = h->d_ru; break;
        case LJMD_V: base = h->d_v; break;
        case LJMD_A: base = h->d_a; break;
        default: return nullptr;
    }
    return base + (size_t)axis * h->P;
}

void *ljmd_stream(ljmd_t *h) { return (h && !h->multi) ? (void *)h->stream : nullptr; }

int ljmd_step_begin(ljmd_t *h)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_step_begin: NULL handle");
    if (h->multi) return fail(h, LJMD_ERR_STATE, "ljmd_step_begin: a multi-device handle runs the exchange phases itself");
    if (!h->have_state || !h->have_accel)
        return fail(h, LJMD_ERR_STATE, "ljmd_step_begin: state/accelerations not initialised");
    LJMD_HIP(h, hipSetDevice(h->device));
    return enqueue_drift(h, next_events(h));
}

int ljmd_step_forces(ljmd_t *h)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_step_forces: NULL handle");
    if (h->multi) return fail(h, LJMD_ERR_STATE, "ljmd_step_forces: a multi-device handle runs the exchange phases itself");
    if (!h->have_state) return fail(h, LJMD_ERR_STATE, "ljmd_step_forces: no state has been set");
    LJMD_HIP(h, hipSetDevice(h->device));
    // pairs with the event set taken by ljmd_step_begin (the last one handed out)
    EventSet *q = (h->profiling && h->ev_used > 0) ? &h->ev_pool[h->ev_used - 1] : nullptr;
    return enqueue_pair_forces(h, q);
}

int ljmd_step_finish(ljmd_t *h)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_step_finish: NULL handle");
    if (h->multi) return fail(h, LJMD_ERR_STATE, "ljmd_step_finish: a multi-device handle runs the exchange phases itself");
    if (!h->have_state) return fail(h, LJMD_ERR_STATE, "ljmd_step_finish: no state has been set");
    LJMD_HIP(h, hipSetDevice(h->device));
    EventSet *q = (h->profiling && h->ev_used > 0) ? &h->ev_pool[h->ev_used - 1] : nullptr;
    if (!h->forces_pending) {
        const int rc_ = enqueue_pair_forces(h, q);
        if (rc_ != LJMD_OK) return rc_;
    }
    return enqueue_kick(h, true, q);
}

int ljmd_force_buffers(ljmd_t *h, int32_t external, void **fpart, int64_t *fpart_doubles, void **frecv,
                       int64_t *frecv_doubles)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_force_buffers: NULL handle");
    if (h->multi) return fail(h, LJMD_ERR_STATE, "ljmd_force_buffers: a multi-device handle runs the exchange phases itself");
    h->external_force_exchange = external != 0;
    if (fpart) *fpart = h->d_fpart;
    if (fpart_doubles) *fpart_doubles = 3 * (int64_t)h->P * (needs_force_exchange(h) ? h->G : 1);
    if (frecv) *frecv = h->d_frecv;
    if (frecv_doubles) *frecv_doubles = needs_force_exchange(h) ? 3 * (int64_t)h->P : 0;
    return LJMD_OK;
}

int ljmd_forces_partial(ljmd_t *h)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_forces_partial: NULL handle");
    if (h->multi) return fail(h, LJMD_ERR_STATE, "ljmd_forces_partial: a multi-device handle runs the exchange phases itself");
    if (!h->have_state) return fail(h, LJMD_ERR_STATE, "ljmd_forces_partial: no state has been set");
    LJMD_HIP(h, hipSetDevice(h->device));
    if (!h->forces_pending) {
        const int rc_ = enqueue_pair_forces(h, nullptr);
        if (rc_ != LJMD_OK) return rc_;
    }
    return enqueue_kick(h, false, nullptr);
}

int ljmd_read_partials(ljmd_t *h, int32_t nsteps, double *partial)
{
    if (!h || !partial || nsteps < 0 || nsteps > (int)kRingCap)
        return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_read_partials: bad argument");
    if (h->multi) return fail(h, LJMD_ERR_STATE, "ljmd_read_partials: a multi-device handle combines its ranks itself");
    LJMD_HIP(h, hipSetDevice(h->device));
    int rc_ = fetch_ring(h, (unsigned)nsteps);
    if (rc_ != LJMD_OK) return rc_;
    std::memcpy(partial, h->h_ring, (size_t)nsteps * kPartialStride * sizeof(double));
    return LJMD_OK;
}

int ljmd_combine_scalars(const ljmd_t *h, const double *partials_by_rank, int32_t n_ranks, double *epot,
                         double *ekin, double *d_epot, double *dd_epot)
{
    if (!h || !partials_by_rank || n_ranks < 1)
        return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_combine_scalars: bad argument");
    combine_one(h, partials_by_rank, n_ranks, epot, ekin, d_epot, dd_epot);
    return LJMD_OK;
}

// ---- RCCL exchange (one process per GPU) -------------------------------------

int ljmd_comm_unique_id(char *id_out)
{
    if (!id_out) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_comm_unique_id: NULL buffer");
    static_assert(sizeof(ncclUniqueId) == LJMD_COMM_ID_BYTES, "LJMD_COMM_ID_BYTES out of sync with RCCL");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, LJMD_ERR_HIP, "ncclGetUniqueId failed: %s", ncclGetErrorString(r));
    std::memcpy(id_out, id.internal, sizeof id.internal);
    return LJMD_OK;
}

int ljmd_comm_init(ljmd_t *h, const char *id)
{
    if (!h || !id) return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_comm_init: NULL argument");
    if (h->multi) return fail(h, LJMD_ERR_STATE, "ljmd_comm_init: a multi-device handle owns its communicators");
    if (h->comm) return fail(h, LJMD_ERR_STATE, "ljmd_comm_init: communicator already initialised");
    LJMD_HIP(h, hipSetDevice(h->device));
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, sizeof uid.internal);
    const ncclResult_t r = ncclCommInitRank(&h->comm, h->G, uid, h->rank);
    if (r != ncclSuccess) {
        h->comm = nullptr;
        return fail(h, LJMD_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", h->rank, h->G, ncclGetErrorString(r));
    }
    h->overlap_exchange = env_int("LJMD_OVERLAP_EXCHANGE", 1) != 0;
    LJMD_HIP(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    LJMD_HIP(h, hipEventCreateWithFlags(&h->ev_pos_ready, hipEventDisableTiming));
    LJMD_HIP(h, hipEventCreateWithFlags(&h->ev_gather_done, hipEventDisableTiming));
    return LJMD_OK;
}

// ---- ownership migration (multi-GPU) -------------------------------------------------------------------------------

int ljmd_migrate_pack(ljmd_t *h)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_migrate_pack: NULL handle");
    if (h->multi) return fail(h, LJMD_ERR_STATE, "ljmd_migrate_pack: a multi-device handle migrates through ljmd_migrate");
    if (!h->have_state) return fail(h, LJMD_ERR_STATE, "ljmd_migrate_pack: no state has been set");
    if (h->forces_pending) return fail(h, LJMD_ERR_STATE, "ljmd_migrate_pack: a step is half enqueued");
    LJMD_HIP(h, hipSetDevice(h->device));
    return migrate_pack(h);
}

void *ljmd_migrate_buffer(ljmd_t *h, int64_t *n_total, int64_t *own_off, int64_t *own_cnt)
{
    if (!h || h->multi) return nullptr;
    const int64_t blk = (int64_t)kMigrateRows * h->P;
    if (n_total) *n_total = blk * h->G;
    if (own_off) *own_off = blk * h->rank;
    if (own_cnt) *own_cnt = blk;
    return migrate_buffer(h);
}

int ljmd_migrate_deal(ljmd_t *h)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_migrate_deal: NULL handle");
    if (h->multi) return fail(h, LJMD_ERR_STATE, "ljmd_migrate_deal: a multi-device handle migrates through ljmd_migrate");
    LJMD_HIP(h, hipSetDevice(h->device));
    return migrate_deal(h);
}

int ljmd_migrate(ljmd_t *h)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_migrate: NULL handle");
    if (h->poisoned) return fail(h, LJMD_ERR_STATE, "ljmd_migrate: handle poisoned by an earlier failure; call ljmd_set_state");
    if (h->multi) return ljmdm::migrate_now(h);
    if (h->G == 1) return LJMD_OK;                       // one rank owns everything
    if (!h->comm) return fail(h, LJMD_ERR_STATE, "ljmd_migrate: no communicator (use ljmd_migrate_pack / _deal around your own exchange)");
    int rc_ = ljmd_migrate_pack(h);
    if (rc_ != LJMD_OK) return rc_;
    // all collectives of the communicator on ONE stream (see comm_begin): the blocks of everybody's ru, v, a and ids
    const bool cs = use_comm_stream(h);
    const hipStream_t xs = cs ? h->comm_stream : h->stream;
    if (cs && (rc_ = comm_begin(h)) != LJMD_OK) return rc_;
    const size_t blk = (size_t)kMigrateRows * h->P;
    const ncclResult_t r = ncclAllGather(h->d_mig + (size_t)h->rank * blk, h->d_mig, blk, ncclDouble, h->comm, xs);
    if (r != ncclSuccess) return fail(h, LJMD_ERR_HIP, "ncclAllGather (migration) failed: %s", ncclGetErrorString(r));
    if (cs && (rc_ = comm_end(h)) != LJMD_OK) return rc_;
    if ((rc_ = migrate_deal(h)) != LJMD_OK) return rc_;
    h->gather_done_for_step = false;
    return ljmd_allgather_positions(h);                  // every rank's block changed
}

int ljmd_particle_ids(ljmd_t *h, int32_t *ids)
{
    if (!h || !ids) return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_particle_ids: NULL argument");
    if (h->multi) {                                      // global arrays in the caller's order, whatever migrated inside
        for (int32_t k = 0; k < h->n; ++k) ids[k] = k;
        return LJMD_OK;
    }
    if (!h->migrated) {
        for (int32_t j = 0; j < h->S; ++j) ids[j] = h->rank * h->S + j;
        return LJMD_OK;
    }
    static_assert(sizeof(int32_t) == sizeof(int), "particle ids are int32");
    std::memcpy(ids, h->h_gid0.data(), (size_t)h->S * sizeof(int32_t));
    return LJMD_OK;
}

int ljmd_set_tail_corrections(ljmd_t *h, int32_t on)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_set_tail_corrections: NULL handle");
    // host-side only: the constants are added when the step records are combined (combine_one), on the handle the
    // caller holds -- for a multi-device handle that is the parent
    h->tail_on = on != 0;
    return LJMD_OK;
}

int32_t ljmd_multi_migrations(const ljmd_t *h)
{
    return !h ? 0 : h->multi ? ljmdm::migrations(h) : h->migrations;
}

int32_t ljmd_comm_size(const ljmd_t *h)
{
    if (h && h->multi) return ljmdm::comm_size(h);
    if (!h || !h->comm) return 0;
    int count = 0;
    if (ncclCommCount(h->comm, &count) != ncclSuccess) return 0;
    return count;
}

int ljmd_allgather_positions(ljmd_t *h)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_allgather_positions: NULL handle");
    if (h->multi) return fail(h, LJMD_ERR_STATE, "ljmd_allgather_positions: a multi-device handle runs the exchange phases itself");
    if (h->G == 1 && !h->force_collectives) return LJMD_OK;
    if (!h->comm) return fail(h, LJMD_ERR_STATE, "ljmd_allgather_positions: call ljmd_comm_init first");
    LJMD_HIP(h, hipSetDevice(h->device));
    if (h->gather_done_for_step) {       // already issued by ljmd_step_begin on the communication stream
        h->gather_done_for_step = false;
        return LJMD_OK;
    }
    // serial form (t = 0, re-sort steps, LJMD_OVERLAP_EXCHANGE=0): behind the drift/kick (and re-sort) kernels
    // and ahead of the pair kernel
    EventSet *q = (h->profiling && h->ev_used > 0) ? &h->ev_pool[h->ev_used - 1] : nullptr;
    const bool cs = use_comm_stream(h);
    const hipStream_t xs = cs ? h->comm_stream : h->stream;
    int rc_ = cs ? comm_begin(h) : LJMD_OK;
    if (rc_ != LJMD_OK) return rc_;
    if (q) LJMD_HIP(h, hipEventRecord(q->e[5], xs));
    rc_ = allgather_on(h, xs);
    if (rc_ != LJMD_OK) return rc_;
    if (q) {
        LJMD_HIP(h, hipEventRecord(q->e[6], xs));
        q->has_pos_x = true;
    }
    return cs ? comm_end(h) : LJMD_OK;
}

int ljmd_memcpy(ljmd_t *h, void *dst, const void *src, int64_t bytes, int32_t kind)
{
    if (!h || !dst || !src || bytes < 0 || kind < 1 || kind > 3)
        return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_memcpy: bad argument");
    if (h->multi) return fail(h, LJMD_ERR_STATE, "ljmd_memcpy: not available on a multi-device handle");
    LJMD_HIP(h, hipSetDevice(h->device));
    const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost
                                                                           : hipMemcpyDeviceToDevice;
    LJMD_HIP(h, hipMemcpyAsync(dst, src, (size_t)bytes, k, h->stream));
    LJMD_HIP(h, hipStreamSynchronize(h->stream));
    return LJMD_OK;
}

int ljmd_synchronize(ljmd_t *h)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_synchronize: NULL handle");
    if (h->multi) return ljmdm::synchronize(h);
    LJMD_HIP(h, hipSetDevice(h->device));
    LJMD_HIP(h, hipStreamSynchronize(h->stream));
    LJMD_HIP(h, hipDeviceSynchronize());
    return LJMD_OK;
}

// ---- measurement ---------------------------------------------------------------

const char *ljmd_pair_kernel_name(const ljmd_t *h)
{
    if (!h) return "";
    if (h->multi) return ljmdm::pair_kernel_name(h);
    if (!fast_path_ok(h)) return "pair_rows_generic_kernel";
    if (h->use_n3 && h->mode == LJMD_PRECISION_FP32_FORCE) return "pair_n3_f32_kernel";
    return h->use_n3 ? "pair_n3_kernel" : "pair_tiles_kernel";
}

int ljmd_profile_enable(ljmd_t *h, int32_t on)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_profile_enable: NULL handle");
    if (h->multi) return ljmdm::profile_enable(h, on);
    h->profiling = on != 0;
    h->ev_used = 0;
    return LJMD_OK;
}

int ljmd_profile_read(ljmd_t *h, double *ms_avg, int32_t *launches)
{
    return ljmd_profile_read_ex(h, ms_avg, nullptr, launches);
}

namespace {
// intervals 0..3 as documented for ljmd_profile_read; 4 = position exchange, 5 = force exchange (averages over the
// launches that had one; 0 when none did)
int profile_read_full(ljmd_t *h, double *ms_avg /* [6] */, double *ms_min /* [6] */, int32_t *launches,
                      double *ms_median = nullptr /* [6] */)
{
    LJMD_HIP(h, hipSetDevice(h->device));
    LJMD_HIP(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) LJMD_HIP(h, hipStreamSynchronize(h->comm_stream));
    double acc[6] = {0, 0, 0, 0, 0, 0};  // pair kernel, geometry pre-pass, drift/kick, reduce+finalize, exchanges
    double lo[6] = {1e300, 1e300, 1e300, 1e300, 1e300, 1e300};
    size_t cnt_x[2] = {0, 0};
    std::vector<double> all[6];              // per launch, for the medians
    const int from[6] = {2, 1, 0, 3, 5, 7}, to[6] = {3, 2, 1, 4, 6, 8};
    size_t complete = 0;
    for (size_t k = 0; k < h->ev_used; ++k) {
        const EventSet &q = h->ev_pool[k];
        double one[6] = {0, 0, 0, 0, 0, 0};
        bool ok = true;
        for (int c = 0; c < 4 && ok; ++c) {
            float ms = 0.f;   // a set whose step was only half enqueued has unrecorded events: skip it
            ok = hipEventElapsedTime(&ms, q.e[from[c]], q.e[to[c]]) == hipSuccess;
            one[c] = ms;
        }
        if (!ok) {
            (void)hipGetLastError();
            continue;
        }
        const bool have[2] = {q.has_pos_x, q.has_force_x};
        for (int x = 0; x < 2; ++x) {
            float ms = 0.f;
            if (have[x] && hipEventElapsedTime(&ms, q.e[from[4 + x]], q.e[to[4 + x]]) == hipSuccess) {
                acc[4 + x] += ms;
                lo[4 + x] = std::min(lo[4 + x], (double)ms);
                all[4 + x].push_back(ms);
                ++cnt_x[x];
            } else if (have[x]) {
                (void)hipGetLastError();
            }
        }
        for (int c = 0; c < 4; ++c) {
            acc[c] += one[c];
            lo[c] = std::min(lo[c], one[c]);
            all[c].push_back(one[c]);
        }
        ++complete;
    }
    h->ev_used = complete;
    const double cnt = h->ev_used ? (double)h->ev_used : 1.0;
    for (int c = 0; c < 6; ++c) {
        const double div = c < 4 ? cnt : (cnt_x[c - 4] ? (double)cnt_x[c - 4] : 1.0);
        const bool any = c < 4 ? h->ev_used > 0 : cnt_x[c - 4] > 0;
        if (ms_avg) ms_avg[c] = acc[c] / div;
        if (ms_min) ms_min[c] = any ? lo[c] : 0.0;
        if (ms_median) {
            std::vector<double> &v = all[c];
            std::sort(v.begin(), v.end());
            const size_t m = v.size();
            ms_median[c] = m == 0 ? 0.0 : (m % 2 ? v[m / 2] : 0.5 * (v[m / 2 - 1] + v[m / 2]));
        }
    }
    if (launches) *launches = (int32_t)h->ev_used;
    h->ev_used = 0;
    return LJMD_OK;
}
}  // namespace

int ljmd_profile_read_ex(ljmd_t *h, double *ms_avg, double *ms_min, int32_t *launches)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_profile_read: NULL handle");
    if (h->multi) return ljmdm::profile_read_ex(h, ms_avg, ms_min, launches);
    double a[6], b[6];
    const int rc_ = profile_read_full(h, a, b, launches);
    if (rc_ != LJMD_OK) return rc_;
    if (ms_avg) std::memcpy(ms_avg, a, 4 * sizeof(double));
    if (ms_min) std::memcpy(ms_min, b, 4 * sizeof(double));
    return LJMD_OK;
}

int ljmd_profile_read_rank(ljmd_t *h, int32_t rank, double *ms_avg, double *ms_min, int32_t *launches)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_profile_read_rank: NULL handle");
    if (h->multi) {
        ljmd_t *e = ljmdm::rank_engine(h, rank);
        if (!e) return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_profile_read_rank: rank %d out of range", rank);
        const int rc_ = profile_read_full(e, ms_avg, ms_min, launches);
        if (rc_ != LJMD_OK) return fail(h, rc_, "rank %d (device %d): %s", e->rank, e->device, e->err.c_str());
        return LJMD_OK;
    }
    if (rank != h->rank) return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_profile_read_rank: this engine is rank %d", h->rank);
    return profile_read_full(h, ms_avg, ms_min, launches);
}

int ljmd_profile_read_stats(ljmd_t *h, int32_t rank, double *ms_avg, double *ms_min, double *ms_median, int32_t *launches)
{
    if (!h) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_profile_read_stats: NULL handle");
    if (h->multi) {
        ljmd_t *e = ljmdm::rank_engine(h, rank);
        if (!e) return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_profile_read_stats: rank %d out of range", rank);
        const int rc_ = profile_read_full(e, ms_avg, ms_min, launches, ms_median);
        if (rc_ != LJMD_OK) return fail(h, rc_, "rank %d (device %d): %s", e->rank, e->device, e->err.c_str());
        return LJMD_OK;
    }
    if (rank != h->rank) return fail(h, LJMD_ERR_INVALID_ARG, "ljmd_profile_read_stats: this engine is rank %d", h->rank);
    return profile_read_full(h, ms_avg, ms_min, launches, ms_median);
}

// ---- stateless drop-ins ----------------------------------------------------------

namespace {
std::mutex g_cache_mutex;
ljmd_t *g_cached = nullptr;
bool g_stateless_tail_on = true;         // ljmd_stateless_set_tail_corrections: applied to the cached engine of the drop-ins
// what the last ljmd_verlet_step call handed back (r, v, a; 9 n doubles): if the next call passes exactly these
// bytes again -- the reference's own loop only READS the arrays between steps (md_simulation_program.f90:303-353)
// -- the resident state IS the caller's state and the upload + spatial re-sort can be skipped
std::vector<double> g_last_out;
bool g_last_valid = false;

int cached_engine(int32_t n, double L, double dt, double rc, ljmd_t **out)
{
    if (g_cached && (g_cached->n != n || g_cached->L != L || g_cached->rc != rc)) {
        release(g_cached);
        g_cached = nullptr;
        g_last_valid = false;
    }
    if (!g_cached) {
        int rc_ = ljmd_create(&g_cached, n, L, dt, rc, LJMD_PRECISION_FP64, env_int("LJMD_DEVICE", 0), 0, 1);
        if (rc_ != LJMD_OK) return rc_;
        g_last_valid = false;
    }
    if (g_cached->dt != dt) {
        if (!(dt > 0.0)) return fail(nullptr, LJMD_ERR_INVALID_ARG, "dt must be > 0");
        g_cached->dt = dt;
        g_cached->dt_half = 0.5 * dt;
        g_cached->dt_sq_half = g_cached->dt_half * dt;
    }
    g_cached->tail_on = g_stateless_tail_on;
    *out = g_cached;
    return LJMD_OK;
}

bool same_as_last_output(size_t n, const double *const a[9])
{
    if (!g_last_valid || g_last_out.size() != 9 * n) return false;
    for (int k = 0; k < 9; ++k)
        if (std::memcmp(a[k], g_last_out.data() + (size_t)k * n, n * sizeof(double)) != 0) return false;
    return true;
}
}  // namespace

int ljmd_compute_lj_potential_energy(int32_t n, double box_length, double rc, const double *rx,
                                     const double *ry, const double *rz, double *ax, double *ay,
                                     double *az, double *epot, double *d_epot, double *dd_epot)
{
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    if (!rx || !ry || !rz || !ax || !ay || !az)  // lj_potential_energy.f90:82
        return fail(nullptr, LJMD_ERR_INVALID_ARG, "compute_lj_potential_energy(): state arrays are not allocated.");
    ljmd_t *h = nullptr;
    int rc_ = cached_engine(n, box_length, 1.0, rc, &h);
    if (rc_ != LJMD_OK) return rc_;
    g_last_valid = false;                        // the resident velocities are about to be overwritten with dummies
    // velocities are irrelevant here; reuse the position arrays as dummies
    if ((rc_ = ljmd_set_state(h, rx, ry, rz, rx, ry, rz)) != LJMD_OK) return rc_;
    if ((rc_ = ljmd_compute_forces(h, epot, d_epot, dd_epot)) != LJMD_OK) return rc_;
    rc_ = ljmd_get_state(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                         nullptr, ax, ay, az);
    if (rc_ != LJMD_OK) g_last_error = h->err;
    return rc_;
}

int ljmd_verlet_step(int32_t n, double box_length, double dt, double rc, double *rx, double *ry,
                     double *rz, double *vx, double *vy, double *vz, double *ax, double *ay, double *az,
                     double *epot, double *ekin, double *d_epot, double *dd_epot)
{
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    if (!rx || !ry || !rz || !vx || !vy || !vz || !ax || !ay || !az)  // verlet.f90:52
        return fail(nullptr, LJMD_ERR_INVALID_ARG, "verlet_step(): state arrays are not allocated.");
    ljmd_t *h = nullptr;
    int rc_ = cached_engine(n, box_length, dt, rc, &h);
    if (rc_ != LJMD_OK) return rc_;
    double *const arr[9] = {rx, ry, rz, vx, vy, vz, ax, ay, az};
    const bool resident = env_int("LJMD_STATELESS_FASTPATH", 1) != 0 && !h->poisoned && h->have_state && h->have_accel &&
                          same_as_last_output((size_t)n, arr);
    g_last_valid = false;
    if (!resident) {
        if ((rc_ = ljmd_set_state(h, rx, ry, rz, vx, vy, vz)) != LJMD_OK) return rc_;
        if ((rc_ = ljmd_set_accel(h, ax, ay, az)) != LJMD_OK) return rc_;
    }
    // one step, then the scalar record and the nine arrays behind a single synchronisation
    LJMD_HIP(h, hipSetDevice(h->device));
    rc_ = enqueue_drift(h, nullptr);
    if (rc_ == LJMD_OK) rc_ = enqueue_forces(h, true, nullptr);
    if (rc_ != LJMD_OK) {
        h->poisoned = true;
        g_last_error = h->err;
        return rc_;
    }
    double *const dsts[12] = {rx, ry, rz, nullptr, nullptr, nullptr, vx, vy, vz, ax, ay, az};
    rc_ = download_state(h, dsts, 1);
    if (rc_ != LJMD_OK) {
        g_last_error = h->err;
        return rc_;
    }
    combine_one(h, h->h_ring, 1, epot, ekin, d_epot, dd_epot);
    g_last_out.resize(9 * (size_t)n);
    for (int k = 0; k < 9; ++k) std::memcpy(g_last_out.data() + (size_t)k * n, arr[k], (size_t)n * sizeof(double));
    g_last_valid = true;
    return LJMD_OK;
}

// ---- trajectory analysis: RDF pair pass -------------------------------------------------

int ljmd_rdf_histogram(int32_t n, const double *x, const double *y, const double *z, double box_length,
                       int32_t nbins, double rmax, uint64_t *hist)
{
    if (n < 2 || !x || !y || !z || !hist || nbins < 1 || nbins > 8192 || !(box_length > 0.0) || !(rmax > 0.0))
        return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_rdf_histogram: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, LJMD_ERR_NO_DEVICE, "ljmd_rdf_histogram: no HIP device available (this library has no CPU path)");
    LJMD_HIP(nullptr, hipSetDevice(0));
    double *d = nullptr;
    unsigned long long *dh = nullptr;
    const size_t nb = (size_t)n * sizeof(double);
    auto body = [&]() -> int {
        LJMD_HIP(nullptr, hipMalloc(&d, 3 * nb));
        LJMD_HIP(nullptr, hipMalloc(&dh, (size_t)nbins * sizeof(unsigned long long)));
        LJMD_HIP(nullptr, hipMemcpy(d, x, nb, hipMemcpyHostToDevice));
        LJMD_HIP(nullptr, hipMemcpy(d + n, y, nb, hipMemcpyHostToDevice));
        LJMD_HIP(nullptr, hipMemcpy(d + 2 * (size_t)n, z, nb, hipMemcpyHostToDevice));
        LJMD_HIP(nullptr, hipMemset(dh, 0, (size_t)nbins * sizeof(unsigned long long)));
        RdfArgs a;
        a.x = d;
        a.y = d + n;
        a.z = d + 2 * (size_t)n;
        a.hist = dh;
        a.n = n;
        a.nbins = nbins;
        a.L = box_length;
        a.rmax = rmax;
        a.dr = rmax / nbins;                         // as the reference: dr = rmax / nbins
        a.invL = 1.0 / box_length;
        a.inv_dr = 1.0 / a.dr;
        const int row_blocks = (n + kBlock - 1) / kBlock;
        int ns = std::max(1, std::min((kTargetWorkgroups + row_blocks - 1) / row_blocks, (n + 63) / 64));
        a.chunk = (n + ns - 1) / ns;
        ns = (n + a.chunk - 1) / a.chunk;
        LJMD_HIP(nullptr, launch_rdf_histogram(a, dim3(row_blocks, ns), nullptr));
        std::vector<unsigned long long> hh(nbins);
        LJMD_HIP(nullptr, hipMemcpy(hh.data(), dh, (size_t)nbins * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int b = 0; b < nbins; ++b) hist[b] += hh[b];
        return LJMD_OK;
    };
    const int rc_ = body();
    (void)hipFree(d);
    (void)hipFree(dh);
    return rc_;
}

int ljmd_time_origin_average(int32_t kind, int32_t n_snap, int32_t n, const double *x, const double *y, const double *z,
                             int32_t max_lag, int32_t origin_stride, double *out)
{
    if ((kind != 0 && kind != 1) || n_snap < 2 || n < 1 || !x || !y || !z || !out || max_lag < 0 || origin_stride < 1)
        return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_time_origin_average: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, LJMD_ERR_NO_DEVICE, "ljmd_time_origin_average: no HIP device available (this library has no CPU path)");
    LJMD_HIP(nullptr, hipSetDevice(env_int("LJMD_DEVICE", 0)));
    max_lag = std::min(max_lag, n_snap - 1);
    const int n_origins = (n_snap - 1 + origin_stride - 1) / origin_stride;       // t0 = 0, stride, ... < n_snap - 1
    const size_t bytes = (size_t)n_snap * n * sizeof(double), nterm = (size_t)n_origins * (max_lag + 1);
    double *d = nullptr, *dt = nullptr;
    auto body = [&]() -> int {
        LJMD_HIP(nullptr, hipMalloc(&d, 3 * bytes));
        LJMD_HIP(nullptr, hipMalloc(&dt, nterm * sizeof(double)));
        LJMD_HIP(nullptr, hipMemcpy(d, x, bytes, hipMemcpyHostToDevice));
        LJMD_HIP(nullptr, hipMemcpy(d + (size_t)n_snap * n, y, bytes, hipMemcpyHostToDevice));
        LJMD_HIP(nullptr, hipMemcpy(d + 2 * (size_t)n_snap * n, z, bytes, hipMemcpyHostToDevice));
        TimeOriginArgs a;
        a.x = d;
        a.y = d + (size_t)n_snap * n;
        a.z = d + 2 * (size_t)n_snap * n;
        a.term = dt;
        a.n_snap = n_snap;
        a.n = n;
        a.max_lag = max_lag;
        a.origin_stride = origin_stride;
        LJMD_HIP(nullptr, launch_time_origin(a, kind == 1, n_origins, nullptr));
        std::vector<double> term(nterm);
        LJMD_HIP(nullptr, hipMemcpy(term.data(), dt, nterm * sizeof(double), hipMemcpyDeviceToHost));
        // the reference's accumulation: for t0 ascending, acc[:L + 1] += term(t0, :), counts[:L + 1] += 1, then acc / counts
        std::vector<double> acc(max_lag + 1, 0.0);
        std::vector<long> counts(max_lag + 1, 0);
        for (int k = 0; k < n_origins; ++k) {
            const int t0 = k * origin_stride, L = std::min(max_lag, (n_snap - 1) - t0);
            if (L <= 0) continue;
            for (int lag = 0; lag <= L; ++lag) {
                acc[lag] += term[(size_t)k * (max_lag + 1) + lag];
                counts[lag] += 1;
            }
        }
        for (int lag = 0; lag <= max_lag; ++lag) out[lag] = counts[lag] > 0 ? acc[lag] / (double)counts[lag] : 0.0;
        return LJMD_OK;
    };
    const int rc_ = body();
    (void)hipFree(d);
    (void)hipFree(dt);
    return rc_;
}

void ljmd_stateless_set_tail_corrections(int32_t on)
{
    std::lock_guard<std::mutex> lk(g_cache_mutex);
    g_stateless_tail_on = on != 0;
    if (g_cached) g_cached->tail_on = g_stateless_tail_on;
}

void ljmd_stateless_reset(void)
{
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    if (g_cached) release(g_cached);
    g_cached = nullptr;
    g_last_valid = false;
    std::vector<double>().swap(g_last_out);
}

}  // extern "C"
