// fake_hip_rccl.cpp -- TEST INFRASTRUCTURE: a host-memory stand-in for the HIP runtime and RCCL entry points
// libljmd.so imports, LD_PRELOADed in front of the sanitizer build of the library (csrc/obj/libljmd_asan.so) by
// tests/test_host_orchestration.py.  "Device" memory is calloc'ed host memory, copies are memmove, streams and
// events are inert, kernel launches do nothing, and the RCCL collectives are carried out for real between the
// communicators of one process (ncclCommInitAll + ncclGroupStart/End), so that AddressSanitizer checks every
// extent and offset the host code hands to hipMemcpyAsync / ncclAllGather / ncclReduceScatter / ncclSend+Recv --
// including the single-process multi-GPU path that cannot run on a one-GPU box.  Numbers computed "on the device"
// are meaningless here (no kernel runs); only the host orchestration is under test.  Never linked into the product.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
int fake_devices()
{
    const char *v = std::getenv("FAKEHIP_DEVICES");
    return (v && *v) ? std::atoi(v) : 8;
}
thread_local int t_device = 0;
struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local std::vector<CallCfg> t_cfg;
std::atomic<long> g_launches{0}, g_copies{0}, g_collectives{0};
}  // namespace

extern "C" {

// ---- what the test reads back -----------------------------------------------------------------------------
long fakehip_kernel_launches(void) { return g_launches; }
long fakehip_copies(void) { return g_copies; }
long fakehip_collectives(void) { return g_collectives; }

// ---- device / error ------------------------------------------------------------------------------------------
hipError_t hipGetDeviceCount(int *count) { *count = fake_devices(); return *count > 0 ? hipSuccess : hipErrorNoDevice; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= fake_devices()) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error (fake HIP)" : "error (fake HIP)"; }
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned) { return (peer >= 0 && peer < fake_devices()) ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int)
{
    std::memset(p, 0, sizeof *p);
    std::snprintf(p->name, sizeof p->name, "fake MI355X");
    std::snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->warpSize = 64;
    p->maxThreadsPerBlock = 1024;
    p->maxThreadsDim[0] = p->maxThreadsDim[1] = p->maxThreadsDim[2] = 1024;
    p->maxGridSize[0] = p->maxGridSize[1] = p->maxGridSize[2] = 2147483647;
    p->multiProcessorCount = 256;
    p->sharedMemPerBlock = 65536;
    p->regsPerBlock = 65536;
    p->totalGlobalMem = 288ull << 30;
    p->major = 9;
    p->minor = 5;
    return hipSuccess;
}
int hipGetStreamDeviceId(hipStream_t) { return t_device; }

// ---- memory ----------------------------------------------------------------------------------------------------
hipError_t hipMalloc(void **p, size_t n) { *p = std::calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = std::calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); ++g_copies; return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); ++g_copies; return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }

// ---- streams / events (inert: everything above is synchronous) ------------------------------------------------
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)std::calloc(1, 8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)std::calloc(1, 8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)std::calloc(1, 8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }

// ---- kernel launch plumbing of hipcc's host stubs -----------------------------------------------------------------
void **__hipRegisterFatBinary(const void *) { static void *handle = nullptr; return &handle; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream)
{
    t_cfg.push_back({grid, block, shmem, stream});
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream)
{
    if (t_cfg.empty()) return hipErrorInvalidValue;
    *grid = t_cfg.back().grid; *block = t_cfg.back().block; *shmem = t_cfg.back().shmem; *stream = t_cfg.back().stream;
    t_cfg.pop_back();
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void *, dim3 grid, dim3 block, void **, size_t, hipStream_t)
{
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024)
        return hipErrorInvalidConfiguration;               // what the real runtime rejects, too
    ++g_launches;
    return hipSuccess;
}

}  // extern "C"

// ---- RCCL: real data movement between the communicators of ONE process ---------------------------------------------
struct FakeGroup { int n; };
struct ncclComm { int rank, nranks; FakeGroup *group; };

namespace {
enum OpKind { kAllGather, kReduceScatter, kSend, kRecv };
struct Op { OpKind kind; ncclComm *comm; const void *send; void *recv; size_t count; int peer; };
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

ncclResult_t run_ops(std::vector<Op> &ops)
{
    // collectives: every op needs its partners (same group, same kind, same count) among the queued ops
    for (const Op &o : ops) {
        if (o.kind == kAllGather || o.kind == kReduceScatter) {
            std::vector<const Op *> part(o.comm->nranks, nullptr);
            for (const Op &p : ops)
                if (p.kind == o.kind && p.comm->group == o.comm->group && p.count == o.count) part[p.comm->rank] = &p;
            for (const Op *p : part)
                if (!p) return ncclInvalidUsage;           // a rank is missing: real RCCL would hang
        }
    }
    // all-gather in place (send = own slice of recv) is legal: stage the sends first
    for (const Op &o : ops) {
        if (o.kind != kAllGather) continue;
        for (const Op &p : ops)
            if (p.kind == kAllGather && p.comm->group == o.comm->group && p.comm != o.comm)
                std::memmove((char *)o.recv + (size_t)p.comm->rank * o.count * 8, p.send, o.count * 8);
        std::memmove((char *)o.recv + (size_t)o.comm->rank * o.count * 8, o.send, o.count * 8);
        ++g_collectives;
    }
    for (const Op &o : ops) {
        if (o.kind != kReduceScatter) continue;
        std::vector<double> acc(o.count, 0.0);
        for (int q = 0; q < o.comm->nranks; ++q)           // rank order
            for (const Op &p : ops)
                if (p.kind == kReduceScatter && p.comm->group == o.comm->group && p.comm->rank == q) {
                    const double *s = (const double *)p.send + (size_t)o.comm->rank * o.count;
                    for (size_t i = 0; i < o.count; ++i) acc[i] += s[i];
                }
        std::memcpy(o.recv, acc.data(), o.count * 8);
        ++g_collectives;
    }
    for (const Op &r : ops) {
        if (r.kind != kRecv) continue;
        const Op *match = nullptr;
        for (const Op &s : ops)
            if (s.kind == kSend && s.comm->group == r.comm->group && s.comm->rank == r.peer && s.peer == r.comm->rank &&
                s.count == r.count)
                match = &s;
        if (!match) return ncclInvalidUsage;
        std::memmove(r.recv, match->send, r.count * 8);
        ++g_collectives;
    }
    ops.clear();
    return ncclSuccess;
}

// one host thread per rank (the library's StepTeam): every thread calls its own communicator outside any group, and the
// collective completes when the last rank has arrived -- a rendezvous across the threads of this process
std::mutex g_mu;
std::condition_variable g_cv;
std::vector<Op> g_waiting;
unsigned long g_completed = 0;                               // collectives carried out through the rendezvous

ncclResult_t rendezvous(const Op &o)
{
    std::unique_lock<std::mutex> lk(g_mu);
    g_waiting.push_back(o);
    std::vector<Op> mine;
    for (const Op &p : g_waiting)
        if (p.kind == o.kind && p.comm->group == o.comm->group) mine.push_back(p);
    if ((int)mine.size() < o.comm->nranks) {
        const unsigned long seen = g_completed;
        // (a rank that never arrives would hang real RCCL as well; the test's own timeout ends such a run)
        g_cv.wait(lk, [&] {
            if (g_completed == seen) return false;
            for (const Op &p : g_waiting)
                if (p.comm == o.comm && p.kind == o.kind) return false;      // still queued: another group completed
            return true;
        });
        return ncclSuccess;
    }
    std::vector<Op> rest;
    for (const Op &p : g_waiting)
        if (!(p.kind == o.kind && p.comm->group == o.comm->group)) rest.push_back(p);
    g_waiting.swap(rest);
    const ncclResult_t r = run_ops(mine);
    ++g_completed;
    g_cv.notify_all();
    return r;
}

ncclResult_t submit(const Op &o)
{
    if (t_depth == 0 && o.comm->nranks != 1 && (o.kind == kAllGather || o.kind == kReduceScatter)) return rendezvous(o);
    t_ops.push_back(o);
    if (t_depth > 0) return ncclSuccess;
    if (o.comm->nranks != 1) {                               // ungrouped send / recv between ranks: not supported here
        t_ops.clear();
        return ncclInvalidUsage;
    }
    return run_ops(t_ops);
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { std::memset(id, 0x5a, sizeof *id); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId, int rank)
{
    if (nranks != 1 || rank != 0) return ncclUnhandledCudaError;   // other processes do not exist in the fake
    *comm = new ncclComm{0, 1, new FakeGroup{1}};
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    for (int a = 0; a < ndev; ++a)
        for (int b = a + 1; b < ndev; ++b)
            if (devlist && devlist[a] == devlist[b]) return ncclInvalidUsage;   // RCCL refuses duplicate devices
    FakeGroup *g = new FakeGroup{ndev};
    for (int r = 0; r < ndev; ++r) comms[r] = new ncclComm{r, ndev, g};
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (c && --c->group->n == 0) delete c->group;
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int *count) { *count = c->nranks; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error (fake RCCL)" : "error (fake RCCL)"; }
ncclResult_t ncclGroupStart(void) { ++t_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void)
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    return run_ops(t_ops);
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t)
{
    if (t != ncclDouble) return ncclInvalidArgument;
    return submit({kAllGather, c, send, recv, count, -1});
}
ncclResult_t ncclReduceScatter(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c,
                               hipStream_t)
{
    if (t != ncclDouble || op != ncclSum) return ncclInvalidArgument;
    return submit({kReduceScatter, c, send, recv, count, -1});
}
ncclResult_t ncclSend(const void *send, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t)
{
    if (t != ncclDouble || peer < 0 || peer >= c->nranks) return ncclInvalidArgument;
    return submit({kSend, c, send, nullptr, count, peer});
}
ncclResult_t ncclRecv(void *recv, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t)
{
    if (t != ncclDouble || peer < 0 || peer >= c->nranks) return ncclInvalidArgument;
    return submit({kRecv, c, nullptr, recv, count, peer});
}

}  // extern "C"
