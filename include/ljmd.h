/*
 * ljmd.h -- C ABI of libljmd.so: the MI355X (gfx950) drop-in for the reference's
 * Lennard-Jones force/energy + velocity-Verlet hot path.
 *
 * The reference (Ledicia/Molecular-Dynamics-Simulation---Lennard-Jones-monoatomic-fluid)
 * has no FFI layer; its operator boundary is two Fortran module procedures
 *     compute_lj_potential_energy(params, state, epot, d_epot, dd_epot)
 *                                        scripts/physics/lj_potential_energy.f90:46
 *     verlet_step(params, state, epot, ekin, d_epot, dd_epot)
 *                                        scripts/physics/verlet.f90:41
 * plus the caller-side per-step unwrapped-coordinate update
 *                                        scripts/md_simulation_program.f90:339-353.
 * Derived types with allocatable components are not bind(C)-interoperable, so the
 * arrays cross as raw `double*` (c_loc(state%rx) ...) and the scalars by value.
 * INTEGRATION.md shows the ISO_C_BINDING stub that binds each entry point.
 *
 * Conventions
 *   - every function returns LJMD_OK (0) or a negative ljmd_status; the text of the
 *     last error is available from ljmd_last_error() (the Fortran shim turns a
 *     non-zero status into `stop 'ljmd: ...'`, the reference's own convention,
 *     e.g. lj_potential_energy.f90:77-82).
 *   - all arrays are fp64, length n, structure-of-arrays, 0-based in C.
 *   - one handle = one simulation on one GPU; a handle is not thread-safe
 *     (the reference is serial), independent handles may coexist.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry
 *     point fails with LJMD_ERR_NO_DEVICE.
 */
#ifndef LJMD_H
#define LJMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ljmd ljmd_t;

typedef enum ljmd_status {
    LJMD_OK = 0,
    LJMD_ERR_INVALID_ARG = -1, /* a guard of md_types.f90:143-161 / lj_potential_energy.f90:77-82 failed */
    LJMD_ERR_NO_DEVICE = -2,   /* no HIP device / device code cannot run */
    LJMD_ERR_HIP = -3,         /* a HIP runtime call failed */
    LJMD_ERR_STATE = -4,       /* call sequence error (e.g. step before set_state), or a handle poisoned by a
                                  batch of steps that failed half-way: ljmd_set_state makes it usable again */
    LJMD_ERR_ALLOC = -5
} ljmd_status;

/* precision_mode for ljmd_create */
#define LJMD_PRECISION_FP64 0       /* all arithmetic fp64 (BASELINE configs 1-4)            */
#define LJMD_PRECISION_FP32_FORCE 1 /* mixed: pairs of tiles farther apart than r_split (env
                                       LJMD_FP32_SPLIT, default 5 sigma; 0 = all but the own row group) in
                                       fp32 tile-relative arithmetic, nearer pairs in fp64; fp64 accumulation
                                       and integrator.  Needs n >= 16384.  BASELINE config 5.             */

/* Which state array: argument of ljmd_device_ptr / selectors of get_state. */
enum { LJMD_R = 0, LJMD_RU = 1, LJMD_V = 2, LJMD_A = 3 };
#define LJMD_PARTIAL_STRIDE 8

/* ---- library-level ------------------------------------------------------ */

/* "ljmd <version> gfx950"; never NULL. */
const char *ljmd_version(void);
/* Number of visible HIP devices (0 when there is none); never fails. */
int32_t ljmd_device_count(void);
/* Text of the most recent error on this handle (h == NULL: the most recent
 * error of a failed ljmd_create or of a stateless call on this thread). */
const char *ljmd_last_error(const ljmd_t *h);

/* ---- handle lifecycle --------------------------------------------------- */

/*
 * Creates an engine for n particles in a cubic box.  Replaces init_params +
 * compute_derived_params + init_state (scripts/base/md_types.f90:105-201): the
 * derived constants 1/L, L**3, rc*rc, 0.5*dt, (0.5*dt)*dt are computed on the
 * host exactly as written there, and the same guards apply (n > 0, L > 0,
 * rc > 0, rc < L/2, dt > 0) -> LJMD_ERR_INVALID_ARG.
 * rank/n_ranks select the contiguous particle shard [rank*n/n_ranks,
 * (rank+1)*n/n_ranks) this engine integrates and owns pair rows for (SURVEY
 * 8(e)); n_ranks = 1 is the ordinary single-GPU engine.  n must be divisible
 * by n_ranks.
 */
int ljmd_create(ljmd_t **out, int32_t n, double box_length, double dt, double rc,
                int32_t precision_mode, int32_t device, int32_t rank, int32_t n_ranks);
void ljmd_destroy(ljmd_t *h);

/*
 * ONE host process, n_gpus devices (the thin Fortran driver's way to BASELINE config 4: N = 1 048 576 sharded
 * over the 8 GPUs of a node; the reference's caller loop md_simulation_program.f90:300-391 stays as it is).
 * The returned handle is used with the SAME entry points as a single-GPU handle -- ljmd_set_state / set_accel /
 * set_unwrapped / get_state, ljmd_compute_forces, ljmd_verlet_steps, ljmd_enqueue_steps / collect_steps,
 * ljmd_snapshot_begin / end, ljmd_kinetic_energy, ljmd_synchronize, ljmd_profile_*, ljmd_destroy -- with global
 * (length-n) arrays; inside, rank g = one engine on devices[g] (NULL = devices 0 .. n_gpus-1) integrating particles
 * [g n/n_gpus, (g+1) n/n_gpus), and per step one all-gather of the position blocks and (Newton-3) one
 * reduce-scatter of the partial accelerations, stream-ordered with no host synchronisation: RCCL over xGMI
 * (ncclCommInitAll; the collectives of all ranks grouped from the one host thread), or peer-to-peer copies +
 * a rank-ordered sum when LJMD_MULTI_EXCHANGE=copy or a device is listed more than once (RCCL refuses two ranks
 * on one device; this is how several ranks are rehearsed on one card).  The split-phase functions below
 * (ljmd_step_begin ...) are for the one-process-per-GPU form and return LJMD_ERR_STATE on such a handle.
 * n must be divisible by n_gpus.
 */
int ljmd_create_multi(ljmd_t **out, int32_t n, double box_length, double dt, double rc,
                      int32_t precision_mode, int32_t n_gpus, const int32_t *devices);

/* ---- state transfer ------------------------------------------------------ */

/* Host -> HBM.  Replaces read_rv_init + `ru <- r` (md_simulation_program.f90:221-231).
 * Accelerations are zeroed (init_state semantics). All six pointers required. */
int ljmd_set_state(ljmd_t *h, const double *rx, const double *ry, const double *rz,
                   const double *vx, const double *vy, const double *vz);
/* Host -> HBM for a / ru, so that a caller can resume from a full rva.dat snapshot or
 * run verlet_step on caller-owned accelerations (strict drop-in mode). NULL = keep. */
int ljmd_set_accel(ljmd_t *h, const double *ax, const double *ay, const double *az);
int ljmd_set_unwrapped(ljmd_t *h, const double *ux, const double *uy, const double *uz);
/* HBM -> host; any pointer may be NULL (skipped).  r, ru, v, a as in the four
 * records of an rva.dat snapshot (md_simulation_program.f90:384-387). */
int ljmd_get_state(ljmd_t *h, double *rx, double *ry, double *rz,
                   double *ux, double *uy, double *uz,
                   double *vx, double *vy, double *vz,
                   double *ax, double *ay, double *az);

/* ---- the hot path -------------------------------------------------------- */

/*
 * = compute_lj_potential_energy (lj_potential_energy.f90:46-225) on the resident
 * positions: overwrites the resident accelerations, returns epot, d_epot, dd_epot
 * including the x4 / x24 prefactors (:188-193) and the tail corrections (:205-223).
 */
int ljmd_compute_forces(ljmd_t *h, double *epot, double *d_epot, double *dd_epot);

/*
 * = nsteps x { verlet_step (verlet.f90:41-97) ; unwrapped update
 * (md_simulation_program.f90:339-353) } with no host synchronisation inside.
 * Requires valid resident accelerations (ljmd_compute_forces or ljmd_set_accel
 * first, as the reference's drivers do at md_simulation_program.f90:236).
 * epot/ekin/d_epot/dd_epot: each NULL or an array of nsteps doubles receiving the
 * value after every step.  With epot, d_epot and dd_epot all NULL nobody reads the
 * potential-energy sums and the steps run the forces-only pair kernel (see
 * ljmd_enqueue_steps_sampled); the trajectory is bit-for-bit the same.
 */
int ljmd_verlet_steps(ljmd_t *h, int32_t nsteps,
                      double *epot, double *ekin, double *d_epot, double *dd_epot);

/*
 * Asynchronous form of the production loop (md_simulation_program.f90:300-391), so that the
 * host's snapshot I/O (:374-387) overlaps the GPU's next steps instead of stalling them:
 *
 *   ljmd_enqueue_steps(h, k)      k Verlet steps on the engine's stream; returns at once.
 *                                 At most LJMD_MAX_PENDING_STEPS steps may be pending.
 *   ljmd_collect_steps(h, k, ..)  waits for the engine's stream and returns the scalars of the
 *                                 last k enqueued steps (arrays of k doubles or NULL).
 *   ljmd_snapshot_begin(h)        stream-ordered copy of r, ru, v, a (and the slot order) into
 *                                 a device snapshot buffer -- a few microseconds behind the
 *                                 steps enqueued so far -- then HBM -> pinned host on a SECOND
 *                                 stream; returns at once.  Steps enqueued afterwards run
 *                                 concurrently with that transfer and do not alter the snapshot.
 *   ljmd_snapshot_end(h, ...)     waits for the transfer only (not for the engine's stream) and
 *                                 delivers the twelve arrays as ljmd_get_state does.
 * One snapshot may be in flight at a time (LJMD_ERR_STATE otherwise).
 */
#define LJMD_MAX_PENDING_STEPS 4096
int ljmd_enqueue_steps(ljmd_t *h, int32_t nsteps);
/* As ljmd_enqueue_steps, for a segment of which only the LAST step is sampled -- the reference reads epot, d_epot,
 * dd_epot only where mod(step, output_interval) == 0 (md_simulation_program.f90:361; output_interval = 100 in the
 * reference's input file) although lj_potential_energy.f90 sums them on every call.  The pair kernel of the other
 * nsteps - 1 steps leaves the two energy sums out (forces-only instantiation, -6 % kernel time at n = 262144);
 * positions, velocities, accelerations and ekin are bit-for-bit those of ljmd_enqueue_steps, and ljmd_collect_steps
 * returns NaN for epot, d_epot, dd_epot of the steps that were not sampled. */
int ljmd_enqueue_steps_sampled(ljmd_t *h, int32_t nsteps);
/* The same switch for the phase API of a sharded engine (ljmd_step_begin / ljmd_step_forces / ljmd_step_finish):
 * on = 0 makes the following force evaluations forces-only until it is switched on again (default on).
 * ljmd_compute_forces always evaluates the sums. */
int ljmd_set_observables(ljmd_t *h, int32_t on);
int ljmd_collect_steps(ljmd_t *h, int32_t nsteps,
                       double *epot, double *ekin, double *d_epot, double *dd_epot);
int ljmd_snapshot_begin(ljmd_t *h);
int ljmd_snapshot_end(ljmd_t *h, double *rx, double *ry, double *rz,
                      double *ux, double *uy, double *uz,
                      double *vx, double *vy, double *vz,
                      double *ax, double *ay, double *az);

/* Kinetic energy of the resident velocities, one fused sum as at
 * md_simulation_program.f90:238-240 (t = 0 only). */
int ljmd_kinetic_energy(ljmd_t *h, double *ekin);

/* ---- stateless drop-ins (what the Fortran shim modules bind) ------------- */

/*
 * Exact signature-level replacement of compute_lj_potential_energy: host arrays in,
 * host arrays out, the params fields passed by value.  Internally keeps one cached
 * engine per (n, L, rc) on device 0.
 */
int ljmd_compute_lj_potential_energy(int32_t n, double box_length, double rc,
                                     const double *rx, const double *ry, const double *rz,
                                     double *ax, double *ay, double *az,
                                     double *epot, double *d_epot, double *dd_epot);
/* Exact replacement of verlet_step: the nine state arrays are updated in place. */
int ljmd_verlet_step(int32_t n, double box_length, double dt, double rc,
                     double *rx, double *ry, double *rz,
                     double *vx, double *vy, double *vz,
                     double *ax, double *ay, double *az,
                     double *epot, double *ekin, double *d_epot, double *dd_epot);
/*
 * Trajectory analysis pair pass: adds the ordered-pair distance histogram of ONE
 * snapshot to hist[nbins] -- the O(n^2) loop of compute_rdf (scripts/md_one_run_analysis.py:556-584:
 * d -= L*rint(d/L), r = sqrt(.), bin = int(r / (rmax/nbins)) for r < rmax; every unordered pair
 * counts 2, exactly as the reference's np.add.at(hist, bins, 2.0)).  Integer counts, bit-exact.
 * Host arrays in; device 0.  nbins <= 8192.
 */
int ljmd_rdf_histogram(int32_t n, const double *x, const double *y, const double *z, double box_length,
                       int32_t nbins, double rmax, uint64_t *hist);
/*
 * Trajectory analysis, time-origin averages: MSD(lag) = < |ru(t0 + lag) - ru(t0)|^2 > (kind 0, arrays =
 * unwrapped positions) or VACF(lag) = < v(t0) . v(t0 + lag) > (kind 1, arrays = velocities) over particles and time
 * origins t0 = 0, origin_stride, ..., exactly as compute_msd_tau_timeorig / compute_vacf_tau_timeorig do
 * (scripts/md_one_run_analysis.py:404-489): per (origin, lag) the particle mean on the GPU (the reference's per-element
 * expressions; fixed summation order, equal to numpy's pairwise mean to rounding), the origins added on the host in the
 * reference's order.  x, y, z: [n_snap][n] host arrays; out: [min(max_lag, n_snap - 1) + 1].  n_snap >= 2.
 */
int ljmd_time_origin_average(int32_t kind, int32_t n_snap, int32_t n, const double *x, const double *y, const double *z,
                             int32_t max_lag, int32_t origin_stride, double *out);
/* The stateless entry points keep one cached engine (device LJMD_DEVICE, default 0).  ljmd_verlet_step
 * remembers the nine arrays it handed back; when the next call passes the same bytes again (the reference's
 * loop only reads them between steps) the resident state is stepped directly -- no upload, no spatial re-sort,
 * only the download (LJMD_STATELESS_FASTPATH=0 disables the check).  This frees the cached engine. */
void ljmd_stateless_reset(void);

/* ---- multi-GPU split-phase API (one process per GPU) -------- */

/* [i0, i1) = particle rows this engine owns -- until an ownership migration (ljmd_migrate) after ljmd_set_state: from then on it
 * owns the set ljmd_particle_ids names and this call fails with LJMD_ERR_STATE. */
int ljmd_shard_range(const ljmd_t *h, int32_t *i0, int32_t *i1);
/*
 * Device address of the exchange buffer holding ALL n positions in shard-blocked
 * SoA order: block g (g = 0..n_ranks-1) is x[P] y[P] z[P] of rank g's particles in
 * rank g's current (spatially sorted) slot order, P = n/n_ranks rounded up to a
 * multiple of 256, padding slots = NaN.  Rank g's own block is at offset g*3*P
 * doubles, so one in-place RCCL all-gather of 3*P doubles per rank refreshes it.
 */
void *ljmd_exchange_buffer(ljmd_t *h, int64_t *n_doubles_total, int64_t *own_offset_doubles,
                           int64_t *own_count_doubles);
/* Device address of one resident state array (LJMD_R..LJMD_A, axis 0..2), P slots in
 * the current device slot order; for zero-copy views (e.g. torch via __cuda_array_interface__). */
void *ljmd_device_ptr(ljmd_t *h, int32_t which, int32_t axis);
/*
 * RCCL exchange over xGMI.  Rank 0 obtains an id (ncclGetUniqueId) and ships the
 * LJMD_COMM_ID_BYTES to every rank by any out-of-band channel (bench.py: a gloo
 * broadcast); every rank then calls ljmd_comm_init, which joins the communicator with the
 * rank / n_ranks given to ljmd_create.  ljmd_allgather_positions enqueues ONE in-place
 * ncclAllGather of 3*P doubles per rank on the handle's stream -- ordered behind
 * ljmd_step_begin's kernels and ahead of ljmd_step_finish's, no host synchronisation.
 * With n_ranks = 1 it is a no-op and needs no communicator.
 */
#define LJMD_COMM_ID_BYTES 128
int ljmd_comm_unique_id(char *id_out /* [LJMD_COMM_ID_BYTES] */);
int ljmd_comm_init(ljmd_t *h, const char *id /* [LJMD_COMM_ID_BYTES] */);
int ljmd_allgather_positions(ljmd_t *h);
/* Ranks RCCL itself reports for this handle's communicator (ncclCommCount); 0 = no communicator.
 * bench.py prints it so that a multi-GPU line proves the collective ran over all ranks. */
int32_t ljmd_comm_size(const ljmd_t *h);
/*
 * Ownership migration of a multi-GPU run, on the devices.  A rank owns a fixed SET of particles, which in a liquid
 * diffuses out of the region it filled when it was dealt (about 4 sigma rms in 10 000 steps): the rank's 64-particle
 * tiles grow and the tile-pair test skips less (-9 % step rate over 10 000 steps at n = 65536 on 8 ranks).  A migration
 * deals all n particles out again BY POSITION: every rank packs ru, v, a and the particle ids of its slots, one all-gather
 * (80 n bytes over xGMI; the positions are in the exchange buffer already) brings everybody's to every rank, every rank
 * computes the same split of the n particles into G parts of exactly n / G (stable radix sorts on identical input:
 * identical result everywhere; x-slabs by default, near-cubic k-d blocks with LJMD_MIGRATE_DEAL=blocks -- same total work,
 * but only the translation-symmetric slabs give every rank the same share under the pair kernel's ownership rule), keeps
 * part `rank`, re-sorts it into tiles and joins the next position all-gather.  Between two MD steps only (no step half enqueued); pending step records and a snapshot in flight are
 * not affected.  No arithmetic of the path changes: the trajectory differs only by summation order.
 *   ljmd_migrate           everything, collectives included: a multi-device handle (ljmd_create_multi; also done
 *                          automatically every LJMD_MULTI_MIGRATE_EVERY steps, default 2000, and at ljmd_set_state), or a
 *                          rank engine with an RCCL communicator (every rank calls it at the same step); no-op for 1 rank
 *   ljmd_migrate_pack / _buffer / _deal   the phases around a caller-made exchange of the G blocks of the migration
 *                          buffer (block g: 10 P doubles at offset g * 10 P; host-staged fallback, tests); the caller then
 *                          also repeats the position exchange
 *   ljmd_particle_ids      ids[j], j < n / n_ranks: index, in the arrays given to the last ljmd_set_state, of the particle at
 *                          position j of this rank engine's arrays (ljmd_get_state, ljmd_snapshot_end); rank S + j until a
 *                          migration.  ljmd_set_accel / ljmd_set_unwrapped keep taking the GLOBAL arrays in that order.
 *                          (A multi-device handle keeps the caller's order itself: identity.)
 *   ljmd_multi_migrations  migrations done so far on this handle
 */
int ljmd_migrate(ljmd_t *h);
int ljmd_migrate_pack(ljmd_t *h);
void *ljmd_migrate_buffer(ljmd_t *h, int64_t *n_doubles_total, int64_t *own_offset_doubles, int64_t *own_count_doubles);
int ljmd_migrate_deal(ljmd_t *h);
int ljmd_particle_ids(ljmd_t *h, int32_t *ids);
int32_t ljmd_multi_migrations(const ljmd_t *h);
/* Copy on the handle's stream, then wait: kind 1 = host->device, 2 = device->host, 3 = device->device.
 * For callers that stage the exchange / force buffers themselves (host-staged fallback, tests). */
int ljmd_memcpy(ljmd_t *h, void *dst, const void *src, int64_t bytes, int32_t kind);
/* Blocks the host until everything enqueued on the handle's stream (and device) is done. */
int ljmd_synchronize(ljmd_t *h);
/* The HIP stream (hipStream_t) all of this handle's kernels are launched on. */
void *ljmd_stream(ljmd_t *h);
/* Phase 1: drift + wrap + half-kick + unwrapped update of the owned shard; the new
 * positions are written into the own block of the exchange buffer. */
int ljmd_step_begin(ljmd_t *h);
/*
 * Phase 2 (after the all-gather): pair forces of this rank's share of the pair matrix
 * against all n positions; with the Newton-3 kernel on n_ranks > 1 the library then sums
 * the partial accelerations across ranks with ONE ncclReduceScatter of 3*P doubles per rank
 * on the handle's stream; second half-kick; per-rank partial sums appended to the scalar ring.
 */
int ljmd_step_finish(ljmd_t *h);
/* Pair forces only (t = 0 evaluation) on the exchange buffer contents. */
int ljmd_forces_partial(ljmd_t *h);
/*
 * Test / integration hooks.  ljmd_step_forces runs only the pair kernel + slab reduction of
 * phase 2 (ljmd_step_finish / ljmd_forces_partial then continue from there).  ljmd_force_buffers
 * exposes the partial-acceleration buffer fpart ([n_ranks][3][P], block g = contributions to
 * rank g's particles) and the receive buffer frecv ([3][P]); with external != 0 the library
 * skips its own reduce-scatter and expects the caller to have written sum_ranks fpart[g] into
 * rank g's frecv before ljmd_step_finish (used to emulate several ranks on one GPU).
 */
int ljmd_step_forces(ljmd_t *h);
int ljmd_force_buffers(ljmd_t *h, int32_t external, void **fpart, int64_t *fpart_doubles,
                       void **frecv, int64_t *frecv_doubles);
/*
 * Copies out the raw per-rank partial records of the last `nsteps` finished phases,
 * LJMD_PARTIAL_STRIDE (8) doubles each: { sum r^-12, sum r^-6 over this rank's ordered
 * pairs, sum vx^2, sum vy^2, sum vz^2 over its particles, 0, 0, 0 }.  Resets the ring.
 */
int ljmd_read_partials(ljmd_t *h, int32_t nsteps, double *partial);
/* Host-side, deterministic: combines the n_ranks partial records of ONE step (rank
 * order) into epot, ekin, d_epot, dd_epot incl. prefactors and tail corrections. */
int ljmd_combine_scalars(const ljmd_t *h, const double *partials_by_rank, int32_t n_ranks,
                         double *epot, double *ekin, double *d_epot, double *dd_epot);

/*
 * The reference's switch `use_tail_corrections` (scripts/physics/lj_potential_energy.f90:36, a compile-time
 * parameter, .true. as shipped; :205-223): on = 0 leaves the three mean-field tail constants out of epot, d_epot and
 * dd_epot -- of every scalar this handle returns from now on (they are added on the host when the step records are
 * combined; forces never contain them).  ljmd_stateless_set_tail_corrections does the same for the cached engine behind
 * the stateless drop-ins ljmd_compute_lj_potential_energy / ljmd_verlet_step (process-wide, default on); the Fortran shim
 * modules pass their own `use_tail_corrections` parameter through it on every call.
 */
int ljmd_set_tail_corrections(ljmd_t *h, int32_t on);
void ljmd_stateless_set_tail_corrections(int32_t on);

/* ---- measurement --------------------------------------------------------- */

/*
 * Live HIP-event timing on the handle's own stream.  After ljmd_profile_enable(h, 1)
 * every step / force evaluation records events around its kernels; ljmd_profile_read
 * waits for the stream and returns averages per launch in milliseconds:
 *   ms_avg[0] pair-force kernel          ms_avg[1] geometry pre-pass (tile boxes + mask)
 *   ms_avg[2] drift/kick kernel (+ re-sort)   ms_avg[3] slab reduce + kick + finalize
 * *launches = number of launches averaged; the counters are reset.
 */
int ljmd_profile_enable(ljmd_t *h, int32_t on);
/* Name of the pair-force kernel the next force evaluation will launch (for profile matching). */
const char *ljmd_pair_kernel_name(const ljmd_t *h);
int ljmd_profile_read(ljmd_t *h, double *ms_avg /* [4] */, int32_t *launches);
/* Same, plus the minimum over the launches of each interval (ms_min[2] = the drift/kick kernel alone:
 * the steps that also re-sort are longer).  Either array may be NULL. */
int ljmd_profile_read_ex(ljmd_t *h, double *ms_avg /* [4] */, double *ms_min /* [4] */, int32_t *launches);
/* Per rank, with the two exchanges of a multi-GPU step: intervals 0..3 as above, ms[4] = the position
 * all-gather, ms[5] = the force reduce-scatter / all-to-all (HIP events on the stream that carries the collective,
 * from the moment this rank could start it: the wait for the slowest rank is part of it; 0 when the launches had no
 * exchange).  `rank` selects the rank engine of a multi-device handle (ljmd_create_multi); on an ordinary engine it
 * must be the engine's own rank.  Resets that rank's counters. */
int ljmd_profile_read_rank(ljmd_t *h, int32_t rank, double *ms_avg /* [6] */, double *ms_min /* [6] */,
                           int32_t *launches);
/* The same with the median over the launches of each interval: what tells a 1-2 % kernel change from the +-3 % a box
 * differs from the next by (bench.py: roofline.kernel_ms_min / kernel_ms_median).  Any array may be NULL. */
int ljmd_profile_read_stats(ljmd_t *h, int32_t rank, double *ms_avg /* [6] */, double *ms_min /* [6] */,
                            double *ms_median /* [6] */, int32_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* LJMD_H */
