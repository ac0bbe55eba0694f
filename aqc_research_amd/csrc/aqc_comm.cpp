// aqc_comm_*: the one collective layer of the path, bound straight to librccl (RCCL over xGMI) -- no torch, no MPI.
//
// The fidelity/gradient path shards by independent jobs (seeds x horizons, job_executor.py:136-143): one process
// per GPU, no data-path collective.  What crosses GPUs is the final gather of fixed-size result records
// {cost, fidelity, nit, status, theta[T_max]} and, for the optional column-sharded full-unitary AQC objective, one
// all-reduce of 2 (T + 1) doubles per evaluation.  Both are tiny, so the entry points take HOST buffers and stage
// them through a small device buffer owned by the communicator.
//
// Bootstrap as SURVEY 8e prescribes: rank 0 creates the 128-byte unique id (aqc_comm_unique_id) and hands it to the
// other ranks through a file or the environment (the Python side does that); every rank then calls aqc_comm_create.
// librccl is loaded lazily with dlopen, so the core library has no link-time dependency on it and single-GPU users
// never touch it.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>

#include "../../include/aqc_hip.h"
#include "aqc_launch.h"

namespace {

// the few RCCL declarations this file needs (rccl.h: ncclResult_t = int, ncclSuccess = 0, ncclFloat64 = 8, ncclSum = 0)
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
typedef int (*GetUniqueIdFn)(NcclUniqueId*);
typedef int (*CommInitRankFn)(NcclComm*, int, NcclUniqueId, int);
typedef int (*CommDestroyFn)(NcclComm);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, NcclComm, hipStream_t);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);
typedef int (*GetAsyncErrorFn)(NcclComm, int*);
typedef int (*CommAbortFn)(NcclComm);
constexpr int kNcclFloat64 = 8, kNcclSum = 0;

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllGatherFn all_gather = nullptr;
    AllReduceFn all_reduce = nullptr;
    GetErrorStringFn error_string = nullptr;
    GetAsyncErrorFn get_async_error = nullptr;   // optional: a communicator that lost a peer reports it here
    CommAbortFn comm_abort = nullptr;            // optional: tears a communicator down without waiting for its peers
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    if (r.handle || !r.error.empty()) return r;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) { r.error = std::string("cannot load librccl: ") + dlerror(); return r; }
    r.get_unique_id = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
    r.all_gather = (AllGatherFn)dlsym(r.handle, "ncclAllGather");
    r.all_reduce = (AllReduceFn)dlsym(r.handle, "ncclAllReduce");
    r.error_string = (GetErrorStringFn)dlsym(r.handle, "ncclGetErrorString");
    r.get_async_error = (GetAsyncErrorFn)dlsym(r.handle, "ncclCommGetAsyncError");
    r.comm_abort = (CommAbortFn)dlsym(r.handle, "ncclCommAbort");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.all_reduce) {
        r.error = "librccl lacks an expected symbol";
        r.handle = nullptr;
    }
    return r;
}

int comm_fail(const std::string& msg) { return aqc::set_error(msg); }
int nccl_fail(const char* what, int code) {
    Rccl& r = rccl();
    return comm_fail(std::string(what) + " failed: " + (r.error_string ? r.error_string(code) : "RCCL error") + " (" + std::to_string(code) + ")");
}

}  // namespace

struct aqc_comm {
    NcclComm comm = nullptr;
    int nranks = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;
    double* d_buf = nullptr;   // [send | recv]
    double* h_buf = nullptr;   // pinned mirror of d_buf: the ONLY host memory the asynchronous copies touch.  The caller's
                               // (pageable) arrays are read before anything is enqueued and written after the collective
                               // has completed, so a copy can neither block the polling loop below (pageable D2H copies
                               // are host-synchronous) nor land in memory the caller has freed after a time-out
    size_t cap = 0;            // doubles
    bool broken = false;       // a collective failed or timed out: the communicator was aborted, every later call fails at once
};

namespace {

int ensure_cap(aqc_comm* c, size_t doubles) {
    if (doubles <= c->cap) return 0;
    if (c->d_buf && hipFree(c->d_buf) != hipSuccess) return comm_fail("hipFree failed");
    c->d_buf = nullptr;
    if (c->h_buf && hipHostFree(c->h_buf) != hipSuccess) return comm_fail("hipHostFree failed");
    c->h_buf = nullptr;
    c->cap = 0;
    if (hipMalloc((void**)&c->d_buf, doubles * sizeof(double)) != hipSuccess) return comm_fail("hipMalloc of the staging buffer failed");
    if (hipHostMalloc((void**)&c->h_buf, doubles * sizeof(double), hipHostMallocDefault) != hipSuccess)
        return comm_fail("hipHostMalloc of the pinned staging buffer failed");
    c->cap = doubles;
    return 0;
}

// Waits for the collective enqueued on the communicator's stream WITHOUT blocking in the runtime: the stream is polled,
// and so is the communicator's asynchronous error state.  A rank that died leaves its peers inside the collective's
// kernel forever; after AQC_COMM_TIMEOUT_S seconds (default 300) -- or as soon as RCCL reports the failure -- the
// communicator is aborted (ncclCommAbort), marked broken and the call returns an error, so that a job list fails
// instead of hanging (the reference's run_jobs reports failed jobs, job_executor.py:149-159; it never waits on the dead).
double env_seconds(const char* name, double fallback) {
    const char* e = getenv(name);
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : fallback;
}
double comm_timeout_s() { return env_seconds("AQC_COMM_TIMEOUT_S", 300.0); }

int wait_collective(aqc_comm* c, const char* what) {
    Rccl& r = rccl();
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = comm_timeout_s();
    for (unsigned spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return 0;
        std::string why;
        if (q != hipErrorNotReady) why = std::string("stream error: ") + hipGetErrorString(q);
        if (why.empty() && r.get_async_error && (spin & 63u) == 0) {
            int async = 0;
            const int rc = r.get_async_error(c->comm, &async);
            if (rc != 0 || async != 0) why = std::string("RCCL reports an asynchronous error: ") + (r.error_string ? r.error_string(rc != 0 ? rc : async) : "?");
        }
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (why.empty() && waited > limit) why = "no completion after " + std::to_string((int)limit) + " s (AQC_COMM_TIMEOUT_S): a rank has probably died";
        if (!why.empty()) {
            c->broken = true;
            if (r.comm_abort && c->comm) { (void)r.comm_abort(c->comm); c->comm = nullptr; }
            return comm_fail(std::string(what) + " did not complete -- " + why + "; the communicator was aborted");
        }
        if (waited > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(waited > 0.1 ? 1000 : 50));   // spin first: the records are tiny
    }
}

}  // namespace

extern "C" {

int aqc_comm_unique_id(char* out128) {
    if (!out128) return comm_fail("null output");
    Rccl& r = rccl();
    if (!r.handle) return comm_fail(r.error);
    NcclUniqueId id;
    const int rc = r.get_unique_id(&id);
    if (rc != 0) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(out128, id.internal, sizeof id.internal);
    return 0;
}

int aqc_comm_create(const char* id128, int nranks, int rank, int device, aqc_comm** out) {
    if (!out) return comm_fail("out pointer is null");
    *out = nullptr;
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return comm_fail("invalid communicator arguments");
    Rccl& r = rccl();
    if (!r.handle) return comm_fail(r.error);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return comm_fail("no HIP device available: aqc_comm needs one GPU per rank");
    if (device < 0 || device >= ndev) return comm_fail("device " + std::to_string(device) + " out of range (" + std::to_string(ndev) + " visible)");
    if (hipSetDevice(device) != hipSuccess) return comm_fail("hipSetDevice failed");
    aqc_comm* c = new aqc_comm();
    c->nranks = nranks; c->rank = rank; c->device = device;
    NcclUniqueId id;
    memcpy(id.internal, id128, sizeof id.internal);
    // ncclCommInitRank blocks until ALL ranks have joined: a rank that never starts (or that read a stale id) would hang the
    // others for ever.  It runs in a helper thread; this thread waits AQC_COMM_INIT_TIMEOUT_S (default 180 s) for it.  On
    // time-out the call fails and the helper is left behind, detached, with its own copy of everything it touches (the
    // caller is expected to exit: the run has failed).
    struct Init { std::atomic<int> done{0}; int rc = 0; NcclComm comm = nullptr; };
    auto st = std::make_shared<Init>();
    std::thread([st, id, nranks, rank, device, init = r.comm_init_rank]() {
        if (hipSetDevice(device) != hipSuccess) st->rc = -1;
        else st->rc = init(&st->comm, nranks, id, rank);
        st->done.store(1, std::memory_order_release);
    }).detach();
    const double limit = env_seconds("AQC_COMM_INIT_TIMEOUT_S", 180.0);
    const auto t0 = std::chrono::steady_clock::now();
    while (!st->done.load(std::memory_order_acquire)) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            delete c;
            return comm_fail("ncclCommInitRank did not return within " + std::to_string((int)limit) + " s (AQC_COMM_INIT_TIMEOUT_S): rank " +
                             std::to_string(rank) + " of " + std::to_string(nranks) + " is still waiting for its peers -- a rank failed to start, or "
                             "the ranks do not share one unique id");
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    if (st->rc == -1) { delete c; return comm_fail("hipSetDevice failed in the RCCL initialisation thread"); }
    if (st->rc != 0) { delete c; return nccl_fail("ncclCommInitRank", st->rc); }
    c->comm = st->comm;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { r.comm_destroy(c->comm); delete c; return comm_fail("hipStreamCreate failed"); }
    *out = c;
    return 0;
}

int aqc_comm_destroy(aqc_comm* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->stream) { if (!c->broken) (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->h_buf) (void)hipHostFree(c->h_buf);
    if (c->comm) rccl().comm_destroy(c->comm);
    delete c;
    return 0;
}

int aqc_comm_rank(const aqc_comm* c) { return c ? c->rank : -1; }
int aqc_comm_size(const aqc_comm* c) { return c ? c->nranks : -1; }

// recv[r * count + i] = send of rank r; count doubles per rank (host pointers)
int aqc_comm_allgather(aqc_comm* c, const double* send, double* recv, size_t count) {
    if (!c || !send || !recv || count < 1) return comm_fail("invalid all-gather arguments");
    if (c->broken) return comm_fail("the communicator was aborted after a failed collective");
    if (hipSetDevice(c->device) != hipSuccess) return comm_fail("hipSetDevice failed");
    if (ensure_cap(c, count * (size_t)(c->nranks + 1))) return 1;
    double* d_send = c->d_buf;
    double* d_recv = c->d_buf + count;
    double* h_send = c->h_buf;
    double* h_recv = c->h_buf + count;
    memcpy(h_send, send, count * sizeof(double));
    if (hipMemcpyAsync(d_send, h_send, count * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) return comm_fail("H2D copy failed");
    const int rc = rccl().all_gather(d_send, d_recv, count, kNcclFloat64, c->comm, c->stream);
    if (rc != 0) return nccl_fail("ncclAllGather", rc);
    if (hipMemcpyAsync(h_recv, d_recv, count * c->nranks * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return comm_fail("D2H copy failed");
    if (wait_collective(c, "ncclAllGather")) return 1;   // on failure the caller's memory has not been touched
    memcpy(recv, h_recv, count * c->nranks * sizeof(double));
    return 0;
}

// in place: data[i] <- sum over ranks (or max, op = 1) of data[i]
int aqc_comm_allreduce(aqc_comm* c, double* data, size_t count, int op) {
    if (!c || !data || count < 1) return comm_fail("invalid all-reduce arguments");
    if (op != 0 && op != 2) return comm_fail("op must be 0 (sum) or 2 (max)");
    if (c->broken) return comm_fail("the communicator was aborted after a failed collective");
    if (hipSetDevice(c->device) != hipSuccess) return comm_fail("hipSetDevice failed");
    if (ensure_cap(c, count)) return 1;
    memcpy(c->h_buf, data, count * sizeof(double));
    if (hipMemcpyAsync(c->d_buf, c->h_buf, count * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) return comm_fail("H2D copy failed");
    const int rc = rccl().all_reduce(c->d_buf, c->d_buf, count, kNcclFloat64, op == 0 ? kNcclSum : 2 /* ncclMax */, c->comm, c->stream);
    if (rc != 0) return nccl_fail("ncclAllReduce", rc);
    if (hipMemcpyAsync(c->h_buf, c->d_buf, count * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return comm_fail("D2H copy failed");
    if (wait_collective(c, "ncclAllReduce")) return 1;   // on failure the caller's memory has not been touched
    memcpy(data, c->h_buf, count * sizeof(double));
    return 0;
}

int aqc_comm_barrier(aqc_comm* c) {
    double token = 1.0;
    return aqc_comm_allreduce(c, &token, 1, 0);
}

}  // extern "C"
