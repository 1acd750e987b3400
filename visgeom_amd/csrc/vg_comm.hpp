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

#include <rccl/rccl.h>  // types and prototypes only; every call goes through the table below

#include "vg_internal.hpp"

struct vg_comm {
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0, device = 0;
    bool owned = true;  // created here (destroyed here) or adopted from the host
    int replicas = 0;   // > 0: no RCCL behind it -- this rank stands for `replicas` ranks holding identical shards
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

__global__ void vg_scale_in_place_kernel(double *buf, size_t n, double f)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] *= f;
}

// in-place sum over the ranks of c, enqueued on `stream`; a NULL or one-rank communicator is the identity
inline int allreduce_sum(const vg_comm *c, double *device_buf, size_t n, hipStream_t stream)
{
    if (!c || c->n_ranks <= 1 || !n) return VG_OK;
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
    delete c;
}

}  // extern "C"
