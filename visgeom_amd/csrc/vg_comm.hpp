// vg_comm.hpp -- the ONE exchange step of the path, natively: a summing all-reduce (RCCL over xGMI) of the small
// normal-equation blocks, issued on the problem's HIP stream on DEVICE buffers, in place.  Nothing like it exists in the
// reference (single process, single thread; SURVEY section 5 "distributed communication backend: absent"); it serves
// the image-sharded replacement of ceres::Solve at src/calibration/unified_calibration.cpp:53.
//
// RCCL is bound at run time (dlopen + dlsym) so that the one-GPU path neither needs nor loads it ("the
// 1-GPU path must not depend on RCCL"), and so that the copy of librccl the process already uses -- a host application's
// or PyTorch's -- is never interposed by link order.  Messages are a few KB: latency bound, one collective per
// evaluation, no bucketing, no host staging.
#pragma once

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <mutex>

#include <rccl/rccl.h>  // types and prototypes only; every call goes through the table below

#include "vg_internal.hpp"

namespace vgc {
// The ranks of an in-process communicator (vg_comm_create_local): N host threads of ONE process driving ONE device, each
// with its own problem, stream and shard.  A collective is: every rank parks its buffer in its slot, all meet at a host
// barrier, every rank adds the slots in rank order (the same fixed order everywhere: identical totals bit for bit).
struct LocalGroup {
    int n = 0, device = 0, refs = 0;
    size_t cap = 0;            // doubles per slot
    double *slots = nullptr;   // device [n][cap]
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    bool broken = false;       // a rank failed or timed out: every later collective fails instead of hanging
    bool barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const unsigned long long gen = generation;
        if (++arrived == n) {
            arrived = 0;
            generation++;
            cv.notify_all();
            return true;
        }
        // Success is decided by the generation alone: a waiter that was released by the last arrival has completed the
        // collective, also when a faster rank has meanwhile left the group (vg_comm_destroy marks it broken for those who
        // would otherwise wait for the leaver) before this thread re-acquired the mutex.
        cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != gen || broken; });
        if (generation != gen) return true;
        broken = true;  // timed out, or a rank left while this one was still waiting for it
        cv.notify_all();
        return false;
    }
};
}  // namespace vgc

struct vg_comm {
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0, device = 0;
    bool owned = true;  // created here (destroyed here) or adopted from the host
    int replicas = 0;   // > 0: no RCCL behind it -- this rank stands for `replicas` ranks holding identical shards
    vgc::LocalGroup *local = nullptr;  // in-process ranks (threads) on one device, no RCCL behind it
};

namespace vgc {

struct Api {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;
    decltype(&ncclCommUserRank) comm_user_rank = nullptr;
    std::string error;
};

inline Api &api()
{
    static Api a = [] {
        Api t;
        const char *env = getenv("VG_RCCL_LIBRARY");
        const char *names[] = {env, "librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            t.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (t.handle) break;
            t.error = dlerror();
        }
        if (!t.handle) return t;
#define VG_SYM(field, name)                                                      \
    t.field = reinterpret_cast<decltype(t.field)>(dlsym(t.handle, name));        \
    if (!t.field) {                                                              \
        t.error = std::string("symbol ") + name + " missing from the RCCL library"; \
        t.handle = nullptr;                                                      \
        return t;                                                                \
    }
        VG_SYM(get_unique_id, "ncclGetUniqueId")
        VG_SYM(comm_init_rank, "ncclCommInitRank")
        VG_SYM(comm_destroy, "ncclCommDestroy")
        VG_SYM(all_reduce, "ncclAllReduce")
        VG_SYM(error_string, "ncclGetErrorString")
        VG_SYM(comm_count, "ncclCommCount")
        VG_SYM(comm_user_rank, "ncclCommUserRank")
#undef VG_SYM
        return t;
    }();
    return a;
}

inline int need_api()
{
    if (api().handle) return VG_OK;
    return vgi::fail(VG_ERR_STATE, "RCCL is not available: " + api().error);
}

#define VG_NCCL(expr)                                                                                    \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess)                                                                           \
            return vgi::fail(VG_ERR_HIP, std::string(#expr) + ": " + vgc::api().error_string(r_));        \
    } while (0)

#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ void vg_scale_in_place_kernel(double *buf, size_t n, double f)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] *= f;
}
#endif

#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ void vg_local_sum_slots_kernel(const double *slots, size_t cap, int n_ranks, double *buf, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.;
    for (int r = 0; r < n_ranks; r++) s += slots[(size_t)r * cap + i];
    buf[i] = s;
}
#endif

// in-place sum over the ranks of c, enqueued on `stream`; a NULL or one-rank communicator is the identity
inline int allreduce_sum(const vg_comm *c, double *device_buf, size_t n, hipStream_t stream)
{
    if (!c || c->n_ranks <= 1 || !n) return VG_OK;
    if (c->local) {  // in-process ranks: host-synchronous (a test transport, not a fast one)
        LocalGroup *g = c->local;
        // messages longer than a slot (the raw pose blocks of a long coupled sequence, the packed Gram blocks of many
        // datasets) go through the slots piece by piece; every rank walks the same pieces in the same order
        for (size_t off = 0; off < n; off += g->cap) {
            const size_t m = n - off < g->cap ? n - off : g->cap;
            VG_HIP(hipMemcpyAsync(g->slots + (size_t)c->rank * g->cap, device_buf + off, sizeof(double) * m, hipMemcpyDeviceToDevice, stream));
            VG_HIP(hipStreamSynchronize(stream));
            if (!g->barrier()) return vgi::fail(VG_ERR_STATE, "local communicator: a rank failed or did not arrive");
            hipLaunchKernelGGL(vg_local_sum_slots_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream,
                               (const double *)g->slots, g->cap, g->n, device_buf + off, m);
            VG_HIP(hipGetLastError());
            VG_HIP(hipStreamSynchronize(stream));
            if (!g->barrier()) return vgi::fail(VG_ERR_STATE, "local communicator: a rank failed or did not arrive");  // slots free again
        }
        return VG_OK;
    }
    if (c->replicas > 0) {  // the sum over `replicas` identical ranks
        hipLaunchKernelGGL(vg_scale_in_place_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, device_buf, n,
                           (double)c->replicas);
        VG_HIP(hipGetLastError());
        return VG_OK;
    }
    VG_NCCL(api().all_reduce(device_buf, device_buf, n, ncclDouble, ncclSum, c->comm, stream));
    return VG_OK;
}

}  // namespace vgc

extern "C" {

int vg_comm_unique_id(char *id /* VG_COMM_ID_BYTES */)
{
    if (!id) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "id is NULL");
    static_assert(sizeof(ncclUniqueId) == VG_COMM_ID_BYTES, "unique-id size");
    int rc = vgc::need_api();
    if (rc != VG_OK) return rc;
    ncclUniqueId u;
    VG_NCCL(vgc::api().get_unique_id(&u));
    std::memcpy(id, &u, sizeof u);
    return VG_OK;
}

int vg_comm_create(vg_comm **out, const char *id, int n_ranks, int rank, int device)
{
    if (!out || !id) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "rank outside [0, n_ranks)");
    int rc = vgc::need_api();
    if (rc != VG_OK) return rc;
    VG_HIP(hipSetDevice(device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    vg_comm *c = new (std::nothrow) vg_comm();
    if (!c) return vgi::fail(VG_ERR_ALLOC, "out of host memory");
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->device = device;
    ncclResult_t r = vgc::api().comm_init_rank(&c->comm, n_ranks, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return vgi::fail(VG_ERR_HIP, std::string("ncclCommInitRank: ") + vgc::api().error_string(r));
    }
    *out = c;
    return VG_OK;
}

int vg_comm_adopt(vg_comm **out, void *nccl_comm, int device)
{
    if (!out || !nccl_comm) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    int rc = vgc::need_api();
    if (rc != VG_OK) return rc;
    vg_comm *c = new (std::nothrow) vg_comm();
    if (!c) return vgi::fail(VG_ERR_ALLOC, "out of host memory");
    c->comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    c->owned = false;
    c->device = device;
    ncclResult_t r = vgc::api().comm_count(c->comm, &c->n_ranks);
    if (r == ncclSuccess) r = vgc::api().comm_user_rank(c->comm, &c->rank);
    if (r != ncclSuccess) {
        delete c;
        return vgi::fail(VG_ERR_HIP, std::string("ncclCommCount / ncclCommUserRank: ") + vgc::api().error_string(r));
    }
    *out = c;
    return VG_OK;
}

int vg_comm_create_replicated(vg_comm **out, int replicas, int device)
{
    if (!out) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (replicas < 1) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "replicas must be positive");
    vg_comm *c = new (std::nothrow) vg_comm();
    if (!c) return vgi::fail(VG_ERR_ALLOC, "out of host memory");
    c->n_ranks = replicas;
    c->rank = 0;
    c->device = device;
    c->owned = false;
    c->replicas = replicas;
    *out = c;
    return VG_OK;
}

int vg_comm_create_local(vg_comm **out, int n_ranks, int device)
{
    if (!out) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "out is NULL");
    for (int r = 0; r < (n_ranks > 0 ? n_ranks : 0); r++) out[r] = nullptr;
    if (n_ranks < 1 || n_ranks > 64) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "n_ranks must be in [1, 64]");
    VG_HIP(hipSetDevice(device));
    vgc::LocalGroup *g = new (std::nothrow) vgc::LocalGroup();
    if (!g) return vgi::fail(VG_ERR_ALLOC, "out of host memory");
    g->n = n_ranks;
    g->device = device;
    g->cap = 1u << 16;  // 64 Ki doubles per rank and piece; longer messages are walked in pieces (allreduce_sum)
    if (hipMalloc(reinterpret_cast<void **>(&g->slots), sizeof(double) * g->cap * (size_t)n_ranks) != hipSuccess) {
        delete g;
        return vgi::fail(VG_ERR_ALLOC, "hipMalloc of the local communicator's slots failed");
    }
    for (int r = 0; r < n_ranks; r++) {
        vg_comm *c = new (std::nothrow) vg_comm();
        if (!c) {
            for (int q = 0; q < r; q++) { delete out[q]; out[q] = nullptr; }
            (void)hipFree(g->slots);
            delete g;
            return vgi::fail(VG_ERR_ALLOC, "out of host memory");
        }
        c->n_ranks = n_ranks;
        c->rank = r;
        c->device = device;
        c->owned = false;
        c->local = g;
        g->refs++;
        out[r] = c;
    }
    return VG_OK;
}

int vg_comm_size(const vg_comm *c) { return c ? c->n_ranks : -1; }
int vg_comm_rank(const vg_comm *c) { return c ? c->rank : -1; }

int vg_comm_allreduce_sum(vg_comm *c, double *device_buf, int64_t n, void *hip_stream)
{
    if (!c || (n > 0 && !device_buf) || n < 0) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "bad arguments");
    // the explicit entry always goes through RCCL, also with one rank (where the solver skips the call): the binding is
    // exercised on every GPU box, not only on multi-GPU nodes
    if (c->comm && c->n_ranks == 1 && n > 0) {
        VG_NCCL(vgc::api().all_reduce(device_buf, device_buf, (size_t)n, ncclDouble, ncclSum, c->comm,
                                      reinterpret_cast<hipStream_t>(hip_stream)));
        return VG_OK;
    }
    return vgc::allreduce_sum(c, device_buf, (size_t)n, reinterpret_cast<hipStream_t>(hip_stream));
}

void vg_comm_destroy(vg_comm *c)
{
    if (!c) return;
    if (c->owned && c->comm && vgc::api().handle) (void)vgc::api().comm_destroy(c->comm);
    if (c->local) {
        vgc::LocalGroup *g = c->local;
        bool last;
        {
            std::lock_guard<std::mutex> lk(g->m);
            last = --g->refs == 0;
            if (!last) {  // a rank leaving early must not leave the others waiting for it
                g->broken = true;
                g->cv.notify_all();
            }
        }
        if (last) {
            (void)hipSetDevice(g->device);
            (void)hipFree(g->slots);
            delete g;
        }
    }
    delete c;
}

}  // extern "C"
