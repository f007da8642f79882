// ljmd_multi.cpp -- ONE host process driving G MI355X devices (ljmd_create_multi): what a thin Fortran driver
// needs to run BASELINE config 4 (N = 1 048 576 sharded over the 8 GPUs of a node) without a process launcher.
//
// The parent handle owns one ordinary engine per device (rank g of G: it integrates particles [g S, (g+1) S)
// and evaluates those rows of the pair matrix) and issues the two data-path exchanges itself,
// stream-ordered, with no host synchronisation inside a batch of steps:
//   positions   drift kernels of all ranks | all-gather of the 3P-double position blocks | pair kernels
//   forces      (Newton-3) pair kernels + slab reductions | reduce-scatter of fpart[G][3][P] | kick kernels
// exchange = rccl (default for distinct devices): ncclCommInitAll communicators, each collective issued for all
//   ranks inside one ncclGroupStart/End from this one thread;
// exchange = copy (LJMD_MULTI_EXCHANGE=copy, and always when a device is listed twice -- RCCL refuses two ranks
//   on one device, so this is also how several ranks are rehearsed on one card): every rank PULLS the blocks it
//   needs with peer-to-peer hipMemcpyAsync behind the owner's event and adds the G force blocks in rank order
//   (launch_sum_blocks): explicit, run-to-run deterministic summation order;
// exchange = host (LJMD_MULTI_EXCHANGE=host): the same pulls staged through pinned host memory (device -> host by the
//   owner, host -> device by every reader) -- needs neither RCCL nor peer access: the last rung of bench.py's ladder.
// Every exchange runs on a per-rank COMMUNICATION stream, fenced against the rank's engine stream by events, so that the
// position exchange overlaps the velocity half-kick of K1 (verlet.f90:72-74 reads a(t) and v only; the positions are final
// at :58-63) exactly as in the one-process-per-GPU form (ljmd_capi.cpp: enqueue_drift); LJMD_OVERLAP_EXCHANGE=0 puts
// everything back on the engine streams.
// The per-step scalar records stay on the devices; they are read back once per batch and combined on the host
// in rank order (combine_one), exactly as the multi-process path does.
#include "ljmd_multi.h"

#include <atomic>
#include <condition_variable>
#include <memory>
#include <thread>

using namespace ljmdk;
using namespace ljmdh;

enum Exchange { kRccl, kCopy, kHost };

// one host thread per rank for the step loop (ljmd_multi.cpp: team_*)
struct StepTeam {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    unsigned seq = 0;                             // command counter: a new value = a new batch of steps
    bool quit = false, sampled = false;
    int nsteps = 0, done = 0;
    std::vector<int> rc;
    std::atomic<int> arrived{0}, generation{0};   // spinning barrier between the phases of a step
    std::atomic<bool> broken{false};              // a rank failed: nobody waits for it any more
};

struct ljmd_multi {
    int G = 0;
    std::vector<ljmd_t *> eng;
    std::vector<int> dev;
    Exchange xmode = kCopy;
    bool overlap = true;                          // LJMD_OVERLAP_EXCHANGE: exchanges on xs[g] instead of the engine streams
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> xs;                  // per rank: the stream that carries its exchanges
    std::vector<hipStream_t> xs_owned;            // ... those of them this handle created
    // per rank, all without timing: positions final (engine stream) / everything received (xs); partial forces ready
    // (engine stream) / summed forces received (xs); host exchange: own block staged in pinned host memory (xs)
    std::vector<hipEvent_t> ev_pos, ev_got, ev_force, ev_fgot, ev_hpos, ev_hforce;
    std::vector<double *> h_xpos, h_xforce;       // host exchange: pinned staging, [3P] and [G][3P] doubles per rank
    std::vector<double> recs;                     // [G][kPartialStride] scratch of one step
    // Ownership migration.  The ranks own index ranges of the ENGINE order; `owner[k]` = the caller's index of the particle
    // at engine index k (identity until the first migration).  Every `migrate_every` steps the particles are dealt out again
    // by position -- rank g takes the g-th n / G of them along the longest axis -- because a fixed SET of particles diffuses
    // out of the slab it filled at t = 0 and the rank's 64-particle tiles grow (profiles/r02_shard_mixing_long_run.txt).
    std::vector<int32_t> owner;
    std::vector<int32_t> owner_snap;              // the table as of ljmd_snapshot_begin: a snapshot in flight is delivered
                                                  // through it, whatever migrations happen before ljmd_snapshot_end
    int migrate_every = 0, steps_since_migration = 0, migrations = 0;
    std::vector<double> stage[12];                // engine-order staging of r, ru, v, a (x, y, z each)
    std::unique_ptr<StepTeam> team;               // LJMD_MULTI_THREADS (default on for G > 1)
};

namespace ljmdm {

namespace {

#define LJMD_TRY(expr)                      \
    do {                                    \
        const int rc__ = (expr);            \
        if (rc__ != LJMD_OK) return rc__;   \
    } while (0)

// a child's error text becomes the parent's
int child_failed(ljmd_t *h, const ljmd_t *c, int code)
{
    return fail(h, code, "rank %d (device %d): %s", c->rank, c->device, c->err.c_str());
}

#define LJMD_CHILD(h, c, expr)                                   \
    do {                                                         \
        const int rc__ = (expr);                                 \
        if (rc__ != LJMD_OK) return child_failed((h), (c), rc__); \
    } while (0)

// caller order -> engine order (in[] = 3 caller arrays; out = m->stage[base .. base + 2])
void to_engine_order(ljmd_multi *m, int n, const double *const in[3], int base)
{
    for (int a = 0; a < 3; ++a) {
        m->stage[base + a].resize(n);
        for (int k = 0; k < n; ++k) m->stage[base + a][k] = in[a][m->owner[k]];
    }
}

int nccl_failed(ljmd_t *h, const char *what, ncclResult_t r)
{
    return fail(h, LJMD_ERR_HIP, "multi-device %s failed: %s", what, ncclGetErrorString(r));
}

bool exchanges(const ljmd_multi *m) { return m->G > 1 || m->eng[0]->force_collectives; }

// ---------------------------------------------------------------------------------------------------------------------
// The two exchanges, rank by rank.  Every function below is called with rank g's device current and reports errors on
// rank g's own engine handle (so that the ranks can be driven by one host thread each: StepTeam); the caller turns a
// failure into the parent's (child_failed).  An exchange = prepare (what the OWNER of a block does), collect (what the
// RECEIVER does: the collective call / the pulls), finish.  With peer copies or host staging a rank's collect may only
// be issued after EVERY rank's prepare (hipStreamWaitEvent captures the event's latest record at the time of the call):
// the single-thread driver runs the phases as loops over the ranks, the threaded one puts a host barrier between them;
// RCCL needs neither (the collective is its own synchronisation), but from one thread its calls must be grouped.
// ---------------------------------------------------------------------------------------------------------------------

// positions: "rank g's block is final" + (host staging) its copy in pinned memory
int pos_prepare(ljmd_multi *m, int g, EventSet *q)
{
    ljmd_t *e = m->eng[g];
    LJMD_HIP(e, hipEventRecord(m->ev_pos[g], e->stream));
    // the rank's own event first: behind it the rank's previous pair kernel has finished reading the blocks that are
    // about to be overwritten, and the exchange interval (events 5 -> 6) starts when THIS rank could start
    LJMD_HIP(e, hipStreamWaitEvent(m->xs[g], m->ev_pos[g], 0));
    if (q) LJMD_HIP(e, hipEventRecord(q->e[5], m->xs[g]));
    if (m->xmode == kHost) {
        LJMD_HIP(e, hipMemcpyAsync(m->h_xpos[g], own_block(e), 3 * (size_t)e->P * sizeof(double), hipMemcpyDeviceToHost,
                                   m->xs[g]));
        LJMD_HIP(e, hipEventRecord(m->ev_hpos[g], m->xs[g]));
    }
    return LJMD_OK;
}

// positions: every other rank's block into rank d's exchange buffer
int pos_collect(ljmd_multi *m, int d)
{
    ljmd_t *dst = m->eng[d];
    const size_t blk = 3 * (size_t)dst->P;
    if (m->xmode == kRccl) {
        const ncclResult_t r = ncclAllGather(own_block(dst), dst->d_pos, blk, ncclDouble, m->comm[d], m->xs[d]);
        if (r != ncclSuccess) return fail(dst, LJMD_ERR_HIP, "all-gather failed: %s", ncclGetErrorString(r));
        return LJMD_OK;
    }
    for (int g = 0; g < m->G; ++g) {
        if (g == d) continue;
        if (m->xmode == kHost) {
            LJMD_HIP(dst, hipStreamWaitEvent(m->xs[d], m->ev_hpos[g], 0));
            LJMD_HIP(dst, hipMemcpyAsync(dst->d_pos + (size_t)g * blk, m->h_xpos[g], blk * sizeof(double),
                                         hipMemcpyHostToDevice, m->xs[d]));
        } else {
            LJMD_HIP(dst, hipStreamWaitEvent(m->xs[d], m->ev_pos[g], 0));
            LJMD_HIP(dst, hipMemcpyAsync(dst->d_pos + (size_t)g * blk, own_block(m->eng[g]), blk * sizeof(double),
                                         hipMemcpyDeviceToDevice, m->xs[d]));
        }
    }
    return LJMD_OK;
}

int pos_finish(ljmd_multi *m, int g, EventSet *q)
{
    ljmd_t *e = m->eng[g];
    if (q) {
        LJMD_HIP(e, hipEventRecord(q->e[6], m->xs[g]));
        q->has_pos_x = true;
    }
    LJMD_HIP(e, hipEventRecord(m->ev_got[g], m->xs[g]));
    return LJMD_OK;
}

// the engine stream resumes behind the position exchange
int pos_await(ljmd_multi *m, int g)
{
    LJMD_HIP(m->eng[g], hipStreamWaitEvent(m->eng[g]->stream, m->ev_got[g], 0));
    return LJMD_OK;
}

// forces (Newton-3): rank d's frecv = sum over the ranks g of block d of rank g's fpart, in rank order
int force_prepare(ljmd_multi *m, int g, EventSet *q)
{
    ljmd_t *e = m->eng[g];
    LJMD_HIP(e, hipEventRecord(m->ev_force[g], e->stream));
    LJMD_HIP(e, hipStreamWaitEvent(m->xs[g], m->ev_force[g], 0));
    if (q) LJMD_HIP(e, hipEventRecord(q->e[7], m->xs[g]));
    if (m->xmode == kHost) {
        LJMD_HIP(e, hipMemcpyAsync(m->h_xforce[g], e->d_fpart, (size_t)m->G * 3 * e->P * sizeof(double), hipMemcpyDeviceToHost,
                                   m->xs[g]));
        LJMD_HIP(e, hipEventRecord(m->ev_hforce[g], m->xs[g]));
    }
    return LJMD_OK;
}

int force_collect(ljmd_multi *m, int d)
{
    ljmd_t *dst = m->eng[d];
    const size_t blk = 3 * (size_t)dst->P;
    if (m->xmode == kRccl) {
        const ncclResult_t r = ncclReduceScatter(dst->d_fpart, dst->d_frecv, blk, ncclDouble, ncclSum, m->comm[d], m->xs[d]);
        if (r != ncclSuccess) return fail(dst, LJMD_ERR_HIP, "reduce-scatter failed: %s", ncclGetErrorString(r));
        return LJMD_OK;
    }
    for (int g = 0; g < m->G; ++g) {
        if (m->xmode == kHost) {
            LJMD_HIP(dst, hipStreamWaitEvent(m->xs[d], m->ev_hforce[g], 0));
            LJMD_HIP(dst, hipMemcpyAsync(dst->d_fall + (size_t)g * blk, m->h_xforce[g] + (size_t)d * blk, blk * sizeof(double),
                                         hipMemcpyHostToDevice, m->xs[d]));
        } else {
            if (g != d) LJMD_HIP(dst, hipStreamWaitEvent(m->xs[d], m->ev_force[g], 0));
            LJMD_HIP(dst, hipMemcpyAsync(dst->d_fall + (size_t)g * blk, m->eng[g]->d_fpart + (size_t)d * blk,
                                         blk * sizeof(double), hipMemcpyDeviceToDevice, m->xs[d]));
        }
    }
    LJMD_HIP(dst, launch_sum_blocks(dst->d_fall, dst->d_frecv, m->G, (int)blk, m->xs[d]));   // rank order
    return LJMD_OK;
}

int force_finish(ljmd_multi *m, int g, EventSet *q)
{
    ljmd_t *e = m->eng[g];
    if (q) {
        LJMD_HIP(e, hipEventRecord(q->e[8], m->xs[g]));
        q->has_force_x = true;
    }
    LJMD_HIP(e, hipEventRecord(m->ev_fgot[g], m->xs[g]));
    LJMD_HIP(e, hipStreamWaitEvent(e->stream, m->ev_fgot[g], 0));
    return LJMD_OK;
}

// ---- one MD step of rank g in three pieces; the exchanges' collect phases go between them --------------------------
struct RankStep {
    EventSet *q = nullptr;
    bool split = false;
};

// Without a force exchange (gather kernels: small systems, LJMD_N3=0) nothing orders a rank's next drift -- which
// overwrites its position block -- behind the OTHER ranks' pulls of that block in the previous step: with Newton-3 the
// kick waits for everybody's forces, which closes the hazard by itself.
bool needs_pull_guard(const ljmd_multi *m) { return m->G > 1 && m->xmode != kRccl && !needs_force_exchange(m->eng[0]); }

int step_a(ljmd_multi *m, int g, RankStep &st)
{
    ljmd_t *e = m->eng[g];
    if (needs_pull_guard(m))
        for (int d = 0; d < m->G; ++d)
            if (d != g) LJMD_HIP(e, hipStreamWaitEvent(e->stream, m->ev_got[d], 0));
    // K1 in two halves: the positions (verlet.f90:58-63 + the unwrapped update) are final first, their exchange starts
    // on the communication stream and the velocity half-kick (:72-74) runs beside it
    st.q = next_events(e);
    st.split = false;
    if (m->overlap)
        LJMD_TRY(enqueue_drift_positions(e, st.q, &st.split));     // (a re-sort step has already run all of K1)
    else
        LJMD_TRY(enqueue_drift(e, st.q));
    return exchanges(m) ? pos_prepare(m, g, st.q) : LJMD_OK;
}

int step_b(ljmd_multi *m, int g, RankStep &st)
{
    ljmd_t *e = m->eng[g];
    if (exchanges(m)) LJMD_TRY(pos_finish(m, g, st.q));
    if (st.split) LJMD_TRY(enqueue_drift_velocities(e));
    if (exchanges(m)) LJMD_TRY(pos_await(m, g));
    LJMD_TRY(enqueue_pair_forces(e, st.q));
    return needs_force_exchange(e) ? force_prepare(m, g, st.q) : LJMD_OK;
}

int step_c(ljmd_multi *m, int g, RankStep &st, bool kick)
{
    ljmd_t *e = m->eng[g];
    if (needs_force_exchange(e)) LJMD_TRY(force_finish(m, g, st.q));
    return enqueue_kick(e, kick, st.q);
}

// ---- the single-thread driver: every phase as a loop over the ranks ----------------------------------------------------
#define LJMD_RANKS(h, m, expr)                                        \
    for (int g = 0; g < (m)->G; ++g) {                                \
        LJMD_HIP((h), hipSetDevice((m)->dev[g]));                     \
        LJMD_CHILD((h), (m)->eng[g], (expr));                         \
    }

int collect_all(ljmd_t *h, bool forces)
{
    ljmd_multi *m = h->multi;
    ncclResult_t r = m->xmode == kRccl ? ncclGroupStart() : ncclSuccess;
    if (r != ncclSuccess) return nccl_failed(h, "group start", r);
    int rc_ = LJMD_OK;
    for (int g = 0; g < m->G && rc_ == LJMD_OK; ++g) {
        if (hipSetDevice(m->dev[g]) != hipSuccess) rc_ = fail(h, LJMD_ERR_HIP, "hipSetDevice(%d) failed", m->dev[g]);
        if (rc_ == LJMD_OK) {
            const int rg = forces ? force_collect(m, g) : pos_collect(m, g);
            if (rg != LJMD_OK) rc_ = child_failed(h, m->eng[g], rg);
        }
    }
    if (m->xmode == kRccl) {
        r = ncclGroupEnd();
        if (rc_ == LJMD_OK && r != ncclSuccess) rc_ = nccl_failed(h, forces ? "reduce-scatter" : "all-gather", r);
    }
    return rc_;
}

// position exchange outside a step (after set_state / a migration): issued and awaited
int exchange_positions_now(ljmd_t *h)
{
    ljmd_multi *m = h->multi;
    if (!exchanges(m)) return LJMD_OK;
    LJMD_RANKS(h, m, pos_prepare(m, g, nullptr));
    LJMD_TRY(collect_all(h, false));
    LJMD_RANKS(h, m, pos_finish(m, g, nullptr));
    LJMD_RANKS(h, m, pos_await(m, g));
    return LJMD_OK;
}

// forces of all ranks on the positions in the exchange buffers (already exchanged) + optional second half-kick
int enqueue_forces_all(ljmd_t *h, bool kick)
{
    ljmd_multi *m = h->multi;
    const bool fx = needs_force_exchange(m->eng[0]);
    LJMD_RANKS(h, m, enqueue_pair_forces(m->eng[g], nullptr));
    if (fx) {
        LJMD_RANKS(h, m, force_prepare(m, g, nullptr));
        LJMD_TRY(collect_all(h, true));
        LJMD_RANKS(h, m, force_finish(m, g, nullptr));
    }
    LJMD_RANKS(h, m, enqueue_kick(m->eng[g], kick, nullptr));
    return LJMD_OK;
}

int enqueue_one_step(ljmd_t *h)
{
    ljmd_multi *m = h->multi;
    std::vector<RankStep> st(m->G);
    LJMD_RANKS(h, m, step_a(m, g, st[g]));
    if (exchanges(m)) LJMD_TRY(collect_all(h, false));
    LJMD_RANKS(h, m, step_b(m, g, st[g]));
    if (needs_force_exchange(m->eng[0])) LJMD_TRY(collect_all(h, true));
    LJMD_RANKS(h, m, step_c(m, g, st[g], true));
    return LJMD_OK;
}

// ---- the threaded driver: one host thread per rank (StepTeam) ------------------------------------------------------------
// One thread issuing every launch of G devices costs G x (launches + event calls) of host time per step: 1.36 ms at G = 8,
// n = 262144 with peer copies, 46 % of a rank's 2.97 ms step (profiles/r03_multi_handle_host_enqueue_one_thread.txt),
// and more than a whole step for smaller systems.  The team's threads live as long as the handle; a batch of steps is one
// command, inside which thread g issues rank g's launches and ITS side of both exchanges.  Host barriers (spinning: a
// few microseconds) stand where a rank's pulls must not be issued before the other ranks' "block ready" events are
// recorded; with RCCL there is none -- every thread calls its own communicator, the one-device-per-thread usage.
bool team_barrier(ljmd_multi *m)
{
    StepTeam &t = *m->team;
    const int gen = t.generation.load(std::memory_order_acquire);
    if (t.arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == m->G) {
        t.arrived.store(0, std::memory_order_relaxed);
        t.generation.fetch_add(1, std::memory_order_release);
    } else {
        for (int spins = 0; t.generation.load(std::memory_order_acquire) == gen; ++spins) {
            if (t.broken.load(std::memory_order_acquire)) return false;
            if (spins > 4000) std::this_thread::yield();
        }
    }
    return !t.broken.load(std::memory_order_acquire);
}

int team_steps_rank(ljmd_multi *m, int g, int nsteps, bool sampled)
{
    ljmd_t *e = m->eng[g];
    const bool barriers = m->xmode != kRccl && m->G > 1, fx = needs_force_exchange(e), guard = needs_pull_guard(m);
    const bool keep = e->want_energy;
    int rc_ = LJMD_OK;
    auto sync = [&]() { if (rc_ == LJMD_OK && !team_barrier(m)) rc_ = LJMD_ERR_STATE; };   // another rank failed
    for (int s = 0; s < nsteps && rc_ == LJMD_OK; ++s) {
        if (sampled) e->want_energy = s == nsteps - 1;
        RankStep st;
        if (guard) sync();                         // the other ranks' "pulls done" events of the previous step exist
        if (rc_ == LJMD_OK) rc_ = step_a(m, g, st);
        if (barriers) sync();
        if (rc_ == LJMD_OK && exchanges(m)) rc_ = pos_collect(m, g);
        if (rc_ == LJMD_OK) rc_ = step_b(m, g, st);
        if (barriers && fx) sync();
        if (rc_ == LJMD_OK && fx) rc_ = force_collect(m, g);
        if (rc_ == LJMD_OK) rc_ = step_c(m, g, st, true);
    }
    e->want_energy = keep;
    if (rc_ != LJMD_OK) m->team->broken.store(true, std::memory_order_release);   // releases the ranks waiting for this one
    return rc_;
}

void team_worker(ljmd_multi *m, int g)
{
    StepTeam &t = *m->team;
    // a thread that could not select its device would launch on device 0 against another device's buffers
    const hipError_t dev_err = hipSetDevice(m->dev[g]);
    unsigned seen = 0;
    for (;;) {
        int nsteps;
        bool sampled;
        {
            std::unique_lock<std::mutex> lk(t.mu);
            t.cv_go.wait(lk, [&] { return t.seq != seen || t.quit; });
            if (t.quit) return;
            seen = t.seq;
            nsteps = t.nsteps;
            sampled = t.sampled;
        }
        int rc_;
        if (dev_err != hipSuccess) {
            rc_ = fail(m->eng[g], LJMD_ERR_HIP, "rank thread %d: hipSetDevice(%d) failed: %s", g, m->dev[g], hipGetErrorString(dev_err));
            t.broken.store(true, std::memory_order_release);       // releases the ranks that wait for this one
        } else {
            rc_ = team_steps_rank(m, g, nsteps, sampled);
        }
        {
            std::lock_guard<std::mutex> lk(t.mu);
            t.rc[g] = rc_;
            ++t.done;
        }
        t.cv_done.notify_one();
    }
}

int team_enqueue(ljmd_t *h, int nsteps, bool sampled)
{
    ljmd_multi *m = h->multi;
    StepTeam &t = *m->team;
    {
        std::lock_guard<std::mutex> lk(t.mu);
        t.arrived.store(0);
        t.broken.store(false);
        t.nsteps = nsteps;
        t.sampled = sampled;
        t.done = 0;
        ++t.seq;
    }
    t.cv_go.notify_all();
    {
        std::unique_lock<std::mutex> lk(t.mu);
        t.cv_done.wait(lk, [&] { return t.done == m->G; });
    }
    for (int g = 0; g < m->G; ++g)                     // the rank that failed by itself, not the ones it released
        if (t.rc[g] != LJMD_OK && !m->eng[g]->err.empty() && t.rc[g] != LJMD_ERR_STATE) return child_failed(h, m->eng[g], t.rc[g]);
    for (int g = 0; g < m->G; ++g)
        if (t.rc[g] != LJMD_OK) return child_failed(h, m->eng[g], t.rc[g]);
    return LJMD_OK;
}

// reads the last `count` records of every rank and combines them step by step in rank order
int collect(ljmd_t *h, int count, double *epot, double *ekin, double *d_epot, double *dd_epot)
{
    ljmd_multi *m = h->multi;
    for (int g = 0; g < m->G; ++g) {
        LJMD_HIP(h, hipSetDevice(m->dev[g]));
        LJMD_CHILD(h, m->eng[g], fetch_ring(m->eng[g], (unsigned)count));
    }
    m->recs.resize((size_t)m->G * kPartialStride);
    for (int s = 0; s < count; ++s) {
        for (int g = 0; g < m->G; ++g)
            std::memcpy(&m->recs[(size_t)g * kPartialStride], m->eng[g]->h_ring + (size_t)s * kPartialStride,
                        kPartialStride * sizeof(double));
        combine_one(h, m->recs.data(), m->G, epot ? epot + s : nullptr, ekin ? ekin + s : nullptr,
                    d_epot ? d_epot + s : nullptr, dd_epot ? dd_epot + s : nullptr);
    }
    return LJMD_OK;
}

unsigned pending(const ljmd_t *h) { const ljmd_t *e = h->multi->eng[0]; return e->ring_issued - e->ring_consumed; }

}  // namespace

int create(ljmd_t **out, int32_t n, double box_length, double dt, double rc, int32_t precision_mode,
           int32_t n_gpus, const int32_t *devices)
{
    if (!out) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_create_multi: out is NULL");
    *out = nullptr;
    if (n_gpus < 1 || n_gpus > 64) return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_create_multi: n_gpus = %d", n_gpus);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, LJMD_ERR_NO_DEVICE, "ljmd_create_multi: no HIP device available (this library has no CPU path)");
    ljmd_t *h = new (std::nothrow) ljmd;
    ljmd_multi *m = new (std::nothrow) ljmd_multi;
    if (!h || !m) {
        delete h;
        delete m;
        return fail(nullptr, LJMD_ERR_ALLOC, "ljmd_create_multi: out of host memory");
    }
    h->multi = m;
    h->device = -1;
    m->G = n_gpus;
    bool distinct = true;
    for (int g = 0; g < n_gpus; ++g) {
        const int d = devices ? devices[g] : g;
        if (d < 0 || d >= ndev) {
            destroy(h);
            return fail(nullptr, LJMD_ERR_INVALID_ARG, "ljmd_create_multi: device %d for rank %d, but %d device(s) visible", d,
                        g, ndev);
        }
        for (int k : m->dev) distinct = distinct && k != d;
        m->dev.push_back(d);
    }
    const char *xm = std::getenv("LJMD_MULTI_EXCHANGE");
    const std::string want = xm ? xm : "";
    if (!want.empty() && want != "rccl" && want != "copy" && want != "host") {
        destroy(h);
        return fail(nullptr, LJMD_ERR_INVALID_ARG, "LJMD_MULTI_EXCHANGE=%s: expected rccl, copy or host", want.c_str());
    }
    if (want == "rccl" && !distinct) {
        destroy(h);
        return fail(nullptr, LJMD_ERR_INVALID_ARG,
                    "LJMD_MULTI_EXCHANGE=rccl needs %d distinct devices (RCCL refuses two ranks on one device)", n_gpus);
    }
    m->xmode = want == "host" ? kHost : (want == "copy" || !distinct) ? kCopy : kRccl;
    m->overlap = env_int("LJMD_OVERLAP_EXCHANGE", 1) != 0;
    for (int g = 0; g < n_gpus; ++g) {
        ljmd_t *e = nullptr;
        const int rc_ = ljmd_create(&e, n, box_length, dt, rc, precision_mode, m->dev[g], g, n_gpus);
        if (rc_ != LJMD_OK) {
            destroy(h);                                   // g_last_error holds the child's message
            return rc_;
        }
        e->external_force_exchange = true;                // both exchanges are issued here, for all ranks at once
        if (g > 0) e->inject_failure_at = -1;             // fault injection (tests): armed on rank 0 only
        m->eng.push_back(e);
    }
    // the parent carries the parameters the scalar combination needs (tail constants) and what callers query
    const ljmd_t *e0 = m->eng[0];
    h->n = n; h->G = n_gpus; h->S = e0->S; h->P = e0->P; h->mode = precision_mode;
    m->owner.resize(n);
    for (int k = 0; k < n; ++k) m->owner[k] = k;
    // LJMD_MULTI_MIGRATE_EVERY: steps between two ownership migrations (0 = never).  Default 2000: at n = 65536 and 8
    // ranks the step rate falls by 1 % per 1000 steps without it (tools/shard_mixing_probe.py) and one migration costs
    // about as much as 50 steps there
    {
        const char *me = std::getenv("LJMD_MULTI_MIGRATE_EVERY");
        m->migrate_every = me && *me ? std::max(0, std::atoi(me)) : 2000;
    }
    h->L = e0->L; h->invL = e0->invL; h->volume = e0->volume; h->rc = e0->rc; h->rc2 = e0->rc2;
    h->dt = e0->dt; h->dt_half = e0->dt_half; h->dt_sq_half = e0->dt_sq_half;
    h->tail_e = e0->tail_e; h->tail_d = e0->tail_d; h->tail_dd = e0->tail_dd;
    auto body = [&]() -> int {
        std::vector<hipEvent_t> *evs[] = {&m->ev_pos, &m->ev_got, &m->ev_force, &m->ev_fgot, &m->ev_hpos, &m->ev_hforce};
        for (auto *v : evs) v->assign(n_gpus, nullptr);
        m->xs.assign(n_gpus, nullptr);
        m->h_xpos.assign(n_gpus, nullptr);
        m->h_xforce.assign(n_gpus, nullptr);
        for (int g = 0; g < n_gpus; ++g) {
            ljmd_t *e = m->eng[g];
            LJMD_HIP(h, hipSetDevice(m->dev[g]));
            for (auto *v : evs) LJMD_HIP(h, hipEventCreateWithFlags(&(*v)[g], hipEventDisableTiming));
            if (m->overlap) {
                hipStream_t s = nullptr;
                LJMD_HIP(h, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
                m->xs_owned.push_back(s);
                m->xs[g] = s;
            } else {
                m->xs[g] = e->stream;
            }
            if (m->xmode == kCopy)
                for (int k = 0; k < n_gpus; ++k)
                    if (m->dev[k] != m->dev[g]) {
                        const hipError_t pe = hipDeviceEnablePeerAccess(m->dev[k], 0);
                        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                            return fail(h, LJMD_ERR_HIP, "peer access %d -> %d unavailable: %s", m->dev[g], m->dev[k],
                                        hipGetErrorString(pe));
                        (void)hipGetLastError();
                    }
            const size_t blk = 3 * (size_t)e->P * sizeof(double);
            if (m->xmode != kRccl && needs_force_exchange(e) && !e->d_fall) LJMD_HIP(h, hipMalloc(&e->d_fall, blk * n_gpus));
            if (m->xmode == kHost) {
                // portable: every device's copy engine reads the staging of every rank
                LJMD_HIP(h, hipHostMalloc(&m->h_xpos[g], blk, hipHostMallocPortable));
                if (needs_force_exchange(e)) LJMD_HIP(h, hipHostMalloc(&m->h_xforce[g], blk * n_gpus, hipHostMallocPortable));
            }
        }
        if (m->xmode == kRccl) {
            m->comm.assign(n_gpus, nullptr);
            const ncclResult_t r = ncclCommInitAll(m->comm.data(), n_gpus, m->dev.data());
            if (r != ncclSuccess) {
                m->comm.clear();
                return fail(h, LJMD_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", n_gpus, ncclGetErrorString(r));
            }
        }
        if (n_gpus > 1 && env_int("LJMD_MULTI_THREADS", 1) != 0) {
            m->team.reset(new StepTeam);
            m->team->rc.assign(n_gpus, LJMD_OK);
            for (int g = 0; g < n_gpus; ++g) m->team->threads.emplace_back(team_worker, m, g);
        }
        return LJMD_OK;
    };
    const int rc_ = body();
    if (rc_ != LJMD_OK) {
        g_last_error = h->err;
        destroy(h);
        return rc_;
    }
    *out = h;
    return LJMD_OK;
}

void destroy(ljmd_t *h)
{
    if (!h) return;
    ljmd_multi *m = h->multi;
    if (m) {
        if (m->team) {
            {
                std::lock_guard<std::mutex> lk(m->team->mu);
                m->team->quit = true;
            }
            m->team->cv_go.notify_all();
            for (std::thread &t : m->team->threads) t.join();
            m->team.reset();
        }
        for (size_t g = 0; g < m->eng.size(); ++g) {
            (void)hipSetDevice(m->dev[g]);
            if (m->eng[g]->stream) (void)hipStreamSynchronize(m->eng[g]->stream);
        }
        for (hipStream_t x : m->xs_owned) (void)hipStreamSynchronize(x);
        for (ncclComm_t c : m->comm)
            if (c) (void)ncclCommDestroy(c);
        std::vector<hipEvent_t> *evs[] = {&m->ev_pos, &m->ev_got, &m->ev_force, &m->ev_fgot, &m->ev_hpos, &m->ev_hforce};
        for (auto *v : evs)
            for (size_t g = 0; g < v->size(); ++g)
                if ((*v)[g]) {
                    (void)hipSetDevice(m->dev[g]);
                    (void)hipEventDestroy((*v)[g]);
                }
        for (hipStream_t x : m->xs_owned) (void)hipStreamDestroy(x);
        for (double *p : m->h_xpos)
            if (p) (void)hipHostFree(p);
        for (double *p : m->h_xforce)
            if (p) (void)hipHostFree(p);
        for (ljmd_t *e : m->eng) release(e);
        delete m;
    }
    delete h;
}

namespace {
int migrate(ljmd_t *h);
}

int set_state(ljmd_t *h, const double *rx, const double *ry, const double *rz, const double *vx, const double *vy,
              const double *vz)
{
    ljmd_multi *m = h->multi;
    for (int k = 0; k < h->n; ++k) m->owner[k] = k;
    m->steps_since_migration = 0;
    const double *rin[3] = {rx, ry, rz}, *vin[3] = {vx, vy, vz};
    to_engine_order(m, h->n, rin, 0);
    to_engine_order(m, h->n, vin, 6);
    for (ljmd_t *e : m->eng) {
        // after a batch that failed half-way the ranks are a phase apart: every rank drains its stream and
        // re-synchronises its record ring with its own device count (ljmd_set_state on a poisoned engine)
        if (h->poisoned) e->poisoned = true;
        LJMD_CHILD(h, e, ljmd_set_state(e, m->stage[0].data(), m->stage[1].data(), m->stage[2].data(), m->stage[6].data(),
                                        m->stage[7].data(), m->stage[8].data()));
    }
    h->have_state = true;
    h->have_accel = false;
    h->poisoned = false;
    LJMD_TRY(exchange_positions_now(h));      // every rank re-ordered its own block: share the new slot order
    // the first deal is by position too (the caller's order may have nothing to do with space); with migration switched
    // off, or one rank, the ranks own the caller's index ranges
    if (m->G > 1 && m->migrate_every > 0) return migrate(h);
    return LJMD_OK;
}

int set_accel(ljmd_t *h, const double *ax, const double *ay, const double *az)
{
    ljmd_multi *m = h->multi;
    const double *in[3] = {ax, ay, az};
    to_engine_order(m, h->n, in, 9);
    for (ljmd_t *e : m->eng) LJMD_CHILD(h, e, ljmd_set_accel(e, m->stage[9].data(), m->stage[10].data(), m->stage[11].data()));
    h->have_accel = true;
    return LJMD_OK;
}

int set_unwrapped(ljmd_t *h, const double *ux, const double *uy, const double *uz)
{
    ljmd_multi *m = h->multi;
    const double *in[3] = {ux, uy, uz};
    to_engine_order(m, h->n, in, 3);
    for (ljmd_t *e : m->eng) LJMD_CHILD(h, e, ljmd_set_unwrapped(e, m->stage[3].data(), m->stage[4].data(), m->stage[5].data()));
    return LJMD_OK;
}

namespace {
// the children's arrays (engine order, rank blocks) through `fetch`, then out[owner[k]] = engine[k]
template <typename Fetch>
int gather_by_owner(ljmd_t *h, const std::vector<int32_t> &owner, double *const p[12], Fetch &&fetch)
{
    ljmd_multi *m = h->multi;
    for (int k = 0; k < 12; ++k)
        if (p[k]) m->stage[k].resize(h->n);
    for (ljmd_t *e : m->eng) {
        double *q[12];
        for (int k = 0; k < 12; ++k) q[k] = p[k] ? m->stage[k].data() + (size_t)e->rank * e->S : nullptr;
        LJMD_CHILD(h, e, fetch(e, q));
    }
    for (int k = 0; k < 12; ++k)
        if (p[k])
            for (int i = 0; i < h->n; ++i) p[k][owner[i]] = m->stage[k][i];
    return LJMD_OK;
}
}  // namespace

int get_state(ljmd_t *h, double *const p[12])
{
    return gather_by_owner(h, h->multi->owner, p, [](ljmd_t *e, double *const q[12]) {
        return ljmd_get_state(e, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11]);
    });
}

int set_observables(ljmd_t *h, bool on)
{
    ljmd_multi *m = h->multi;
    h->want_energy = on;
    for (int g = 0; g < m->G; ++g) m->eng[g]->want_energy = on;
    return LJMD_OK;
}

int compute_forces(ljmd_t *h, double *epot, double *d_epot, double *dd_epot)
{
    const bool keep = h->want_energy;
    set_observables(h, true);                   // this call exists to return the three sums
    const int rc_ = enqueue_forces_all(h, false);
    set_observables(h, keep);
    if (rc_ != LJMD_OK) return rc_;
    h->have_accel = true;
    return collect(h, 1, epot, nullptr, d_epot, dd_epot);
}

namespace {
// Deal the particles out again by position, on the devices (ljmd.h: ljmd_migrate): every rank packs its ru, v, a and
// ids, the blocks are all-gathered by the handle's exchange (RCCL / peer copies / host staging, on the engine streams:
// a migration is rare and brackets itself with stream synchronisations), every rank computes the same k-d deal and keeps
// its block.  The owner table follows from the ids the ranks report: new engine index g S + j holds the particle that
// had engine index id before, and the ranks' ids start again at the identity (migrate_rebase).
int migrate(ljmd_t *h)
{
    ljmd_multi *m = h->multi;
    const int G = m->G, n = h->n;
    for (int g = 0; g < G; ++g) {
        ljmd_t *e = m->eng[g];
        LJMD_HIP(h, hipSetDevice(m->dev[g]));
        if (e->forces_pending) return fail(h, LJMD_ERR_STATE, "ownership migration: a step is half enqueued");
        LJMD_CHILD(h, e, migrate_pack(e));
    }
    for (int g = 0; g < G; ++g) {        // everything enqueued so far (the exchanges on xs[] included) is done
        LJMD_HIP(h, hipSetDevice(m->dev[g]));
        LJMD_HIP(h, hipStreamSynchronize(m->eng[g]->stream));
        LJMD_HIP(h, hipStreamSynchronize(m->xs[g]));
    }
    const size_t blk = (size_t)kMigrateRows * m->eng[0]->P;
    if (m->xmode == kRccl) {
        ncclResult_t r = ncclGroupStart();
        for (int g = 0; g < G && r == ncclSuccess; ++g) {
            ljmd_t *e = m->eng[g];
            r = ncclAllGather(migrate_buffer(e) + (size_t)g * blk, migrate_buffer(e), blk, ncclDouble, m->comm[g], m->xs[g]);
        }
        const ncclResult_t end = ncclGroupEnd();
        if (r == ncclSuccess) r = end;
        if (r != ncclSuccess) return nccl_failed(h, "all-gather (migration)", r);
    } else {
        std::vector<double *> hs(G, nullptr);
        auto body = [&]() -> int {
            if (m->xmode == kHost)
                for (int g = 0; g < G; ++g) {
                    LJMD_HIP(h, hipSetDevice(m->dev[g]));
                    LJMD_HIP(h, hipHostMalloc(&hs[g], blk * sizeof(double), hipHostMallocPortable));
                    LJMD_HIP(h, hipMemcpy(hs[g], migrate_buffer(m->eng[g]) + (size_t)g * blk, blk * sizeof(double),
                                          hipMemcpyDeviceToHost));
                }
            for (int d = 0; d < G; ++d) {
                LJMD_HIP(h, hipSetDevice(m->dev[d]));
                for (int g = 0; g < G; ++g) {
                    if (g == d) continue;
                    double *dst = migrate_buffer(m->eng[d]) + (size_t)g * blk;
                    if (m->xmode == kHost)
                        LJMD_HIP(h, hipMemcpyAsync(dst, hs[g], blk * sizeof(double), hipMemcpyHostToDevice, m->xs[d]));
                    else
                        LJMD_HIP(h, hipMemcpyAsync(dst, migrate_buffer(m->eng[g]) + (size_t)g * blk, blk * sizeof(double),
                                                   hipMemcpyDeviceToDevice, m->xs[d]));
                }
            }
            for (int g = 0; g < G; ++g) {
                LJMD_HIP(h, hipSetDevice(m->dev[g]));
                LJMD_HIP(h, hipStreamSynchronize(m->xs[g]));
            }
            return LJMD_OK;
        };
        const int rc_ = body();
        for (double *p : hs)
            if (p) (void)hipHostFree(p);
        if (rc_ != LJMD_OK) return rc_;
    }
    for (int g = 0; g < G; ++g) {
        LJMD_HIP(h, hipSetDevice(m->dev[g]));
        LJMD_HIP(h, hipStreamSynchronize(m->xs[g]));
    }
    std::vector<int32_t> owner(n);
    for (int g = 0; g < G; ++g) {
        ljmd_t *e = m->eng[g];
        LJMD_HIP(h, hipSetDevice(m->dev[g]));
        LJMD_CHILD(h, e, migrate_deal(e));                   // synchronises the rank's stream; e->h_gid0 is valid
        for (int j = 0; j < e->S; ++j) {
            const int id = e->h_gid0[j];
            if (id < 0 || id >= n) return fail(h, LJMD_ERR_STATE, "ownership migration: rank %d reports particle id %d", g, id);
            owner[(size_t)g * e->S + j] = m->owner[id];
        }
        LJMD_CHILD(h, e, migrate_rebase(e));
    }
    m->owner.swap(owner);
    m->steps_since_migration = 0;
    ++m->migrations;
    return exchange_positions_now(h);
}
}  // namespace

int migrate_now(ljmd_t *h)
{
    if (!h->have_state) return fail(h, LJMD_ERR_STATE, "ljmd_migrate: no state has been set");
    if (h->multi->G == 1) return LJMD_OK;
    return migrate(h);
}

ljmd_t *rank_engine(ljmd_t *h, int32_t rank)
{
    ljmd_multi *m = h->multi;
    return (rank >= 0 && rank < m->G) ? m->eng[rank] : nullptr;
}

int32_t migrations(const ljmd_t *h) { return h->multi->migrations; }

int enqueue_steps(ljmd_t *h, int32_t nsteps, bool sampled)
{
    {
        ljmd_multi *m = h->multi;
        // between two steps; step records waiting to be collected stay where they are (per-rank sums of finished steps)
        // and a snapshot in flight is delivered through the owner table of its own moment (owner_snap)
        if (m->G > 1 && m->migrate_every > 0 && m->steps_since_migration >= m->migrate_every && h->have_accel)
            LJMD_TRY(migrate(h));
        m->steps_since_migration += nsteps;
    }
    if (pending(h) + (unsigned)nsteps > kRingCap)
        return fail(h, LJMD_ERR_STATE, "ljmd_enqueue_steps: %u + %d pending steps exceed LJMD_MAX_PENDING_STEPS", pending(h),
                    nsteps);
    if (h->multi->team) {
        const int rc_ = team_enqueue(h, nsteps, sampled);
        if (rc_ != LJMD_OK) h->poisoned = true;    // some ranks are a phase ahead of the others
        return rc_;
    }
    const bool keep = h->want_energy;
    for (int s = 0; s < nsteps; ++s) {
        if (sampled) set_observables(h, s == nsteps - 1);
        const int rc_ = enqueue_one_step(h);
        if (rc_ != LJMD_OK) {
            set_observables(h, keep);
            h->poisoned = true;        // some ranks are a phase ahead of the others
            return rc_;
        }
    }
    set_observables(h, keep);
    return LJMD_OK;
}

int collect_steps(ljmd_t *h, int32_t nsteps, double *epot, double *ekin, double *d_epot, double *dd_epot)
{
    return collect(h, nsteps, epot, ekin, d_epot, dd_epot);
}

int snapshot_begin(ljmd_t *h)
{
    for (ljmd_t *e : h->multi->eng) LJMD_CHILD(h, e, ljmd_snapshot_begin(e));
    h->multi->owner_snap = h->multi->owner;
    return LJMD_OK;
}

int snapshot_end(ljmd_t *h, double *const p[12])
{
    // the ranks deliver the engine order of snapshot_begin (their own slot permutation of that moment); migrations since
    // then have changed `owner`, not `owner_snap`
    return gather_by_owner(h, h->multi->owner_snap, p, [](ljmd_t *e, double *const q[12]) {
        return ljmd_snapshot_end(e, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11]);
    });
}

int kinetic_energy(ljmd_t *h, double *ekin)
{
    double total = 0.0;
    for (ljmd_t *e : h->multi->eng) {      // rank order
        double part = 0.0;
        LJMD_CHILD(h, e, ljmd_kinetic_energy(e, &part));
        total += part;
    }
    *ekin = total;
    return LJMD_OK;
}

int synchronize(ljmd_t *h)
{
    for (ljmd_t *e : h->multi->eng) LJMD_CHILD(h, e, ljmd_synchronize(e));
    return LJMD_OK;
}

int profile_enable(ljmd_t *h, int32_t on)
{
    for (ljmd_t *e : h->multi->eng) LJMD_CHILD(h, e, ljmd_profile_enable(e, on));
    return LJMD_OK;
}

int profile_read_ex(ljmd_t *h, double *ms_avg, double *ms_min, int32_t *launches)
{
    // the slowest rank bounds the step: per interval the maximum over the ranks
    double avg[4] = {0, 0, 0, 0}, mn[4] = {0, 0, 0, 0};
    int32_t cnt = 0;
    for (ljmd_t *e : h->multi->eng) {
        double a[4], b[4];
        int32_t c = 0;
        LJMD_CHILD(h, e, ljmd_profile_read_ex(e, a, b, &c));
        for (int k = 0; k < 4; ++k) {
            avg[k] = std::max(avg[k], a[k]);
            mn[k] = std::max(mn[k], b[k]);
        }
        cnt = std::max(cnt, c);
    }
    if (ms_avg) std::memcpy(ms_avg, avg, sizeof avg);
    if (ms_min) std::memcpy(ms_min, mn, sizeof mn);
    if (launches) *launches = cnt;
    return LJMD_OK;
}

const char *pair_kernel_name(const ljmd_t *h) { return ljmd_pair_kernel_name(h->multi->eng[0]); }

int32_t comm_size(const ljmd_t *h)
{
    const ljmd_multi *m = h->multi;
    if (m->xmode != kRccl || m->comm.empty() || !m->comm[0]) return 0;
    int count = 0;
    return ncclCommCount(m->comm[0], &count) == ncclSuccess ? count : 0;
}

}  // namespace ljmdm
