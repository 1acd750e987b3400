// vg_local_impl.hpp -- host side of the localization reprojection costs (vg_local.hpp): resident sets of blocks, the
// batched device entries and the per-block host entries that mirror Ceres' Evaluate contract.  Included at the end of
// vg_capi.hip (the library is one translation unit).
#include <new>

#include "vg_internal.hpp"
#include "vg_local.hpp"

struct vg_reproject_set {
    int device = 0, model = 0, K = 0;
    bool sparse = false;
    hipStream_t stream = nullptr;
    int64_t n_blocks = 0, total = 0;       // total = number of points over all blocks
    std::vector<int64_t> offsets;          // [n_blocks + 1] first point of every block
    double *d_const = nullptr;             // one allocation: intrinsics | xiBaseCam | x1 | x2 | p2 | size
    double *d_intr = nullptr, *d_xb = nullptr, *d_x1 = nullptr, *d_x2 = nullptr, *d_p2 = nullptr, *d_size = nullptr;
    double *d_base = nullptr;              // [kBaseConst] what the frames need of xiBaseCam alone, computed once on the device
    int *d_point_block = nullptr;
    double *d_frames = nullptr;
    // per-block host entry: parameters in, rows out, through one pinned block and one device block
    int64_t max_points = 0;
    double *h_pin = nullptr, *d_io = nullptr;
    // sparse: the most blocks any aligned run of kEmitThreads points of the set touches (counting empty blocks in between)
    int64_t max_wg_span = 0;
};

namespace vgl {

using vgi::fail;

inline void destroy(vg_reproject_set *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->d_const) (void)hipFree(s->d_const);
    if (s->d_point_block) (void)hipFree(s->d_point_block);
    if (s->d_frames) (void)hipFree(s->d_frames);
    if (s->d_io) (void)hipFree(s->d_io);
    if (s->h_pin) (void)hipHostFree(s->h_pin);
    delete s;
}

// io block of the per-block entry: [xiOdom 6 | lengths 5 (mono) | pad 1] then residuals [2 n] then Jacobians
inline size_t io_doubles(const vg_reproject_set *s)
{
    const size_t n = (size_t)s->max_points;
    return 12 + 2 * n + 12 * n + (s->sparse ? 0 : 10 * n);
}

inline int create_unguarded(vg_reproject_set **out, int device, void *hip_stream, int model, const double *intr, const double *xb, bool sparse,
                            int64_t n_blocks, const int64_t *offsets, const double *x1, const double *x2, const double *p2, const double *size,
                            vg_reproject_set *&live)
{
    if (!out) return fail(VG_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    const int K = vg::num_intrinsics(model);
    if (K < 0) return fail(VG_ERR_INVALID_ARGUMENT, "unknown camera model");
    if (!intr || !xb || n_blocks < 0 || (n_blocks > 0 && (!x1 || !p2))) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (sparse && n_blocks > 0 && (!offsets || !x2 || !size)) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_blocks >= (1ll << 31) / 8) return fail(VG_ERR_INVALID_ARGUMENT, "too many blocks");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        (void)hipGetLastError();
        return fail(VG_ERR_NO_DEVICE, "no HIP device: the reprojection costs have no CPU fallback");
    }
    if (device < 0 || device >= n_dev) return fail(VG_ERR_INVALID_ARGUMENT, "device index out of range");
    vg_reproject_set *s = new (std::nothrow) vg_reproject_set();
    if (!s) return fail(VG_ERR_ALLOC, "out of host memory");
    live = s;   // what create() releases if a container below throws
    s->device = device;
    s->model = model;
    s->K = K;
    s->sparse = sparse;
    s->stream = reinterpret_cast<hipStream_t>(hip_stream);
    s->n_blocks = n_blocks;
    s->offsets.resize((size_t)n_blocks + 1);
    for (int64_t b = 0; b <= n_blocks; b++) {
        s->offsets[(size_t)b] = (sparse && n_blocks > 0) ? offsets[b] : b * vg::kMonoPoints;   // an empty sparse set may pass offsets = NULL
        if (b > 0) {
            const int64_t n = s->offsets[(size_t)b] - s->offsets[(size_t)b - 1];
            if (n < 0) {
                delete s;
                live = nullptr;
                return fail(VG_ERR_INVALID_ARGUMENT, "offsets must not decrease");
            }
            s->max_points = n > s->max_points ? n : s->max_points;
        }
    }
    if (n_blocks > 0 && s->offsets[0] != 0) {
        delete s;
        live = nullptr;
        return fail(VG_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
    }
    s->total = s->offsets[(size_t)n_blocks];
    if (s->total >= (1ll << 31)) {
        delete s;
        live = nullptr;
        return fail(VG_ERR_INVALID_ARGUMENT, "too many points");
    }
    const size_t T = (size_t)s->total;
    // layout of the constant block (every piece 16-byte aligned)
    const size_t o_intr = 0, o_xb = 10, o_x1 = 16, o_x2 = o_x1 + 3 * T + (T & 1), o_p2 = o_x2 + (sparse ? 3 * T + (T & 1) : 0),
                 o_size = o_p2 + 2 * T, o_base = o_size + (sparse ? T : 0) + (((sparse ? T : 0)) & 1),
                 n_const = o_base + vg::kBaseConst + 2;
    std::vector<double> h(n_const, 0.);
    std::memcpy(h.data() + o_intr, intr, sizeof(double) * K);
    std::memcpy(h.data() + o_xb, xb, sizeof(double) * 6);
    if (T) {
        std::memcpy(h.data() + o_x1, x1, sizeof(double) * 3 * T);
        if (sparse) std::memcpy(h.data() + o_x2, x2, sizeof(double) * 3 * T);
        std::memcpy(h.data() + o_p2, p2, sizeof(double) * 2 * T);
        if (sparse) std::memcpy(h.data() + o_size, size, sizeof(double) * T);
    }
    int rc = VG_OK;
    auto hip_fail = [&](hipError_t e, const char *what) {
        rc = fail(e == hipErrorNoDevice || e == hipErrorInvalidDevice ? VG_ERR_NO_DEVICE : VG_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
        return rc;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) { hip_fail(e, "hipSetDevice"); destroy(s); live = nullptr; return rc; }
    if ((e = hipMalloc(reinterpret_cast<void **>(&s->d_const), sizeof(double) * n_const)) != hipSuccess) { hip_fail(e, "hipMalloc"); destroy(s); live = nullptr; return rc; }
    if ((e = hipMemcpy(s->d_const, h.data(), sizeof(double) * n_const, hipMemcpyHostToDevice)) != hipSuccess) { hip_fail(e, "hipMemcpy"); destroy(s); live = nullptr; return rc; }
    s->d_intr = s->d_const + o_intr;
    s->d_xb = s->d_const + o_xb;
    s->d_x1 = s->d_const + o_x1;
    s->d_x2 = s->d_const + o_x2;
    s->d_p2 = s->d_const + o_p2;
    s->d_size = s->d_const + o_size;
    s->d_base = s->d_const + o_base;
    hipLaunchKernelGGL(vg::vg_local_base_kernel, dim3(1), dim3(64), 0, s->stream, (const double *)s->d_xb, s->d_base);
    if ((e = hipGetLastError()) != hipSuccess) { hip_fail(e, "vg_local_base_kernel"); destroy(s); live = nullptr; return rc; }
    if (sparse) {
        std::vector<int> pb(T ? T : 1, 0);
        for (int64_t b = 0; b < n_blocks; b++)
            for (int64_t i = s->offsets[(size_t)b]; i < s->offsets[(size_t)b + 1]; i++) pb[(size_t)i] = (int)b;
        // windows of kEmitThreads points starting at ANY point (a launch may start at any block's first point)
        for (size_t i = 0, j = 0; i < T; i++) {
            while (j + 1 < T && j + 1 < i + vg::kEmitThreads) j++;
            const int64_t span = (int64_t)pb[j] - pb[i] + 1;
            s->max_wg_span = span > s->max_wg_span ? span : s->max_wg_span;
        }
        if ((e = hipMalloc(reinterpret_cast<void **>(&s->d_point_block), sizeof(int) * pb.size())) != hipSuccess) { hip_fail(e, "hipMalloc"); destroy(s); live = nullptr; return rc; }
        if ((e = hipMemcpy(s->d_point_block, pb.data(), sizeof(int) * pb.size(), hipMemcpyHostToDevice)) != hipSuccess) { hip_fail(e, "hipMemcpy"); destroy(s); live = nullptr; return rc; }
    }
    const size_t fr = sparse ? (size_t)(n_blocks ? n_blocks : 1) * vg::kSparseFrame : 2;   // mono frames live in LDS only
    if ((e = hipMalloc(reinterpret_cast<void **>(&s->d_frames), sizeof(double) * fr)) != hipSuccess) { hip_fail(e, "hipMalloc"); destroy(s); live = nullptr; return rc; }
    const size_t io = io_doubles(s);
    if ((e = hipMalloc(reinterpret_cast<void **>(&s->d_io), sizeof(double) * io)) != hipSuccess) { hip_fail(e, "hipMalloc"); destroy(s); live = nullptr; return rc; }
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&s->h_pin), sizeof(double) * io, hipHostMallocDefault)) != hipSuccess) { hip_fail(e, "hipHostMalloc"); destroy(s); live = nullptr; return rc; }
    *out = s;
    live = nullptr;
    return VG_OK;
}

// std::vector growth inside the set-up may throw; nothing may cross the extern "C" boundary
inline int create(vg_reproject_set **out, int device, void *hip_stream, int model, const double *intr, const double *xb, bool sparse,
                  int64_t n_blocks, const int64_t *offsets, const double *x1, const double *x2, const double *p2, const double *size)
{
    vg_reproject_set *live = nullptr;
    try {
        return create_unguarded(out, device, hip_stream, model, intr, xb, sparse, n_blocks, offsets, x1, x2, p2, size, live);
    } catch (const std::exception &e) {
        destroy(live);
        if (out) *out = nullptr;
        return fail(VG_ERR_ALLOC, std::string("out of host memory while building the block set: ") + e.what());
    }
}

constexpr size_t kLocalLds = sizeof(double) * (vg::kEmitThreads / vg::kWave) * 2 * vg::kWave * 6;
constexpr unsigned int kFusedMaxGrid = 256;   // one workgroup per CU: beyond that the frame work per workgroup is not hidden

// frames of blocks [b0, b0 + nb) from xi_odom (row r = block b0 + r), then the points [p0, p0 + np)
inline int launch(vg_reproject_set *s, int64_t b0, int64_t nb, const double *xi_odom, const double *lengths, double *res, double *jac0,
                  double *jac1)
{
    if (!nb) return VG_OK;
    VG_HIP(hipSetDevice(s->device));
    const int64_t p0 = s->offsets[(size_t)b0], np = s->offsets[(size_t)(b0 + nb)] - p0;
    const dim3 grid((unsigned)((np + vg::kEmitThreads - 1) / vg::kEmitThreads)), blk(vg::kEmitThreads);
    // few workgroups: each computes the frames of its own blocks (one launch, no frame round trip through HBM)
    const bool fused = s->sparse && grid.x <= kFusedMaxGrid && s->max_wg_span <= vg::kSparseWgBlocks;
    if (s->sparse && !fused) {
        hipLaunchKernelGGL(vg::vg_local_frame_kernel, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, s->stream, (const double *)s->d_xb, (const double *)s->d_base,
                           xi_odom, (long long)b0, (long long)nb, 1, s->d_frames);
        VG_HIP(hipGetLastError());
    }
    if (!np) return VG_OK;
    if (s->sparse) {
        vg::SparseArgs a;
        a.frames = s->d_frames;
        a.intr = s->d_intr;
        a.x1 = s->d_x1;
        a.x2 = s->d_x2;
        a.p2 = s->d_p2;
        a.size = s->d_size;
        a.point_block = s->d_point_block;
        a.xb = s->d_xb;
        a.base = s->d_base;
        a.xi_odom = xi_odom;
        a.first_block = b0;
        a.res = res;
        a.jac = jac0;
        a.first_point = p0;
        a.n_points = (unsigned)np;
        if (fused) {
            const size_t lds = kLocalLds + sizeof(double) * vg::kSparseWgBlocks * vg::kSparseFrame;
            switch (s->model) {
            case vg::kEUCM: hipLaunchKernelGGL((vg::vg_sparse_reproject_kernel<vg::kEUCM, true>), grid, blk, lds, s->stream, a); break;
            case vg::kUCM: hipLaunchKernelGGL((vg::vg_sparse_reproject_kernel<vg::kUCM, true>), grid, blk, lds, s->stream, a); break;
            default: hipLaunchKernelGGL((vg::vg_sparse_reproject_kernel<vg::kMEI, true>), grid, blk, lds, s->stream, a); break;
            }
        } else {
            switch (s->model) {
            case vg::kEUCM: hipLaunchKernelGGL((vg::vg_sparse_reproject_kernel<vg::kEUCM, false>), grid, blk, kLocalLds, s->stream, a); break;
            case vg::kUCM: hipLaunchKernelGGL((vg::vg_sparse_reproject_kernel<vg::kUCM, false>), grid, blk, kLocalLds, s->stream, a); break;
            default: hipLaunchKernelGGL((vg::vg_sparse_reproject_kernel<vg::kMEI, false>), grid, blk, kLocalLds, s->stream, a); break;
            }
        }
    } else {
        vg::MonoArgs a;
        const dim3 mgrid((unsigned)((np + vg::kMonoThreads - 1) / vg::kMonoThreads)), mblk(vg::kMonoThreads);
        a.xb = s->d_xb;
        a.base = s->d_base;
        a.xi_odom = xi_odom;
        a.intr = s->d_intr;
        a.x1 = s->d_x1;
        a.p2 = s->d_p2;
        a.lengths = lengths;
        a.res = res;
        a.jac_odom = jac0;
        a.jac_len = jac1;
        a.first_block = b0;
        a.n_points = (unsigned)np;
        switch (s->model) {
        case vg::kEUCM: hipLaunchKernelGGL((vg::vg_mono_reproject_kernel<vg::kEUCM>), mgrid, mblk, sizeof(double) * vg::kMonoLdsDoubles, s->stream, a); break;
        case vg::kUCM: hipLaunchKernelGGL((vg::vg_mono_reproject_kernel<vg::kUCM>), mgrid, mblk, sizeof(double) * vg::kMonoLdsDoubles, s->stream, a); break;
        default: hipLaunchKernelGGL((vg::vg_mono_reproject_kernel<vg::kMEI>), mgrid, mblk, sizeof(double) * vg::kMonoLdsDoubles, s->stream, a); break;
        }
    }
    VG_HIP(hipGetLastError());
    return VG_OK;
}

// Evaluate of ONE block with Ceres' contract (host pointers): parameters[0] = xiOdom[6] (, parameters[1] = lengths[5]);
// jacobians NULL or an array of 1 (sparse) / 2 (mono) pointers, each NULL or row-major [2 n x block size]
inline int evaluate_host(vg_reproject_set *s, int64_t b, double const *const *parameters, double *residuals, double **jacobians)
{
    if (!s || !parameters || !residuals) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (b < 0 || b >= s->n_blocks) return fail(VG_ERR_INVALID_ARGUMENT, "block index out of range");
    if (!parameters[0] || (!s->sparse && !parameters[1])) return fail(VG_ERR_INVALID_ARGUMENT, "NULL parameter block");
    const size_t n = (size_t)(s->offsets[(size_t)b + 1] - s->offsets[(size_t)b]);
    double *j0 = jacobians ? jacobians[0] : nullptr, *j1 = (jacobians && !s->sparse) ? jacobians[1] : nullptr;
    std::memcpy(s->h_pin, parameters[0], sizeof(double) * 6);
    if (!s->sparse) std::memcpy(s->h_pin + 6, parameters[1], sizeof(double) * 5);
    VG_HIP(hipSetDevice(s->device));
    VG_HIP(hipMemcpyAsync(s->d_io, s->h_pin, sizeof(double) * 12, hipMemcpyHostToDevice, s->stream));
    double *d_res = s->d_io + 12, *d_j0 = d_res + 2 * n, *d_j1 = d_j0 + 12 * n;
    int rc = launch(s, b, 1, s->d_io, s->d_io + 6, d_res, j0 ? d_j0 : nullptr, j1 ? d_j1 : nullptr);
    if (rc != VG_OK) return rc;
    const size_t n_out = 2 * n + (j0 || j1 ? 12 * n : 0) + (j1 ? 10 * n : 0);
    if (n_out) VG_HIP(hipMemcpyAsync(s->h_pin + 12, d_res, sizeof(double) * n_out, hipMemcpyDeviceToHost, s->stream));
    VG_HIP(hipStreamSynchronize(s->stream));
    std::memcpy(residuals, s->h_pin + 12, sizeof(double) * 2 * n);
    if (j0) std::memcpy(j0, s->h_pin + 12 + 2 * n, sizeof(double) * 12 * n);
    if (j1) std::memcpy(j1, s->h_pin + 12 + 14 * n, sizeof(double) * 10 * n);
    return VG_OK;
}

}  // namespace vgl

extern "C" {

int vg_sparse_reproject_create(vg_reproject_set **out, int device, void *hip_stream, int model, const double *intrinsics,
                               const double *xi_base_cam, int64_t n_blocks, const int64_t *offsets, const double *x1, const double *x2,
                               const double *p2, const double *size)
{
    return vgl::create(out, device, hip_stream, model, intrinsics, xi_base_cam, true, n_blocks, offsets, x1, x2, p2, size);
}

int vg_mono_reproject_create(vg_reproject_set **out, int device, void *hip_stream, int model, const double *intrinsics,
                             const double *xi_base_cam, int64_t n_blocks, const double *x1, const double *p2)
{
    return vgl::create(out, device, hip_stream, model, intrinsics, xi_base_cam, false, n_blocks, nullptr, x1, nullptr, p2, nullptr);
}

int64_t vg_reproject_num_blocks(const vg_reproject_set *s) { return s ? s->n_blocks : -1; }
int64_t vg_reproject_num_points(const vg_reproject_set *s) { return s ? s->total : -1; }
int64_t vg_reproject_block_offset(const vg_reproject_set *s, int64_t block)
{
    return (s && block >= 0 && block <= s->n_blocks) ? s->offsets[(size_t)block] : -1;
}

int vg_sparse_reproject_evaluate(vg_reproject_set *s, const double *xi_odom, double *residuals, double *jacobian)
{
    if (!s || !s->sparse) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "not a SparseReprojectCost set");
    if (s->n_blocks && (!xi_odom || !residuals)) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    return vgl::launch(s, 0, s->n_blocks, xi_odom, nullptr, residuals, jacobian, nullptr);
}

int vg_mono_reproject_evaluate(vg_reproject_set *s, const double *xi_odom, const double *lengths, double *residuals, double *jac_odom,
                               double *jac_lengths)
{
    if (!s || s->sparse) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "not a MonoReprojectCost set");
    if (s->n_blocks && (!xi_odom || !lengths || !residuals)) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    return vgl::launch(s, 0, s->n_blocks, xi_odom, lengths, residuals, jac_odom, jac_lengths);
}

int vg_sparse_reproject_block_evaluate(vg_reproject_set *s, int64_t block, double const *const *parameters, double *residuals,
                                       double **jacobians)
{
    if (!s || !s->sparse) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "not a SparseReprojectCost set");
    return vgl::evaluate_host(s, block, parameters, residuals, jacobians);
}

int vg_mono_reproject_block_evaluate(vg_reproject_set *s, int64_t block, double const *const *parameters, double *residuals,
                                     double **jacobians)
{
    if (!s || s->sparse) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "not a MonoReprojectCost set");
    return vgl::evaluate_host(s, block, parameters, residuals, jacobians);
}

int vg_reproject_synchronize(vg_reproject_set *s)
{
    if (!s) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "set is NULL");
    VG_HIP(hipSetDevice(s->device));
    VG_HIP(hipStreamSynchronize(s->stream));
    return VG_OK;
}

void vg_reproject_destroy(vg_reproject_set *s) { vgl::destroy(s); }

int vg_camera_jacobian_evaluate(int device, void *hip_stream, int model, const double *intrinsics, const double *T12, const double *T23,
                                int64_t n, const double *X2, const double *grad, double *dpdxi, double *dfdxi)
{
    const int K = vg::num_intrinsics(model);
    if (K < 0) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "unknown camera model");
    if (!intrinsics || !T12 || n < 0 || (n > 0 && !X2)) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (dfdxi && !grad) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "dfdxi needs the image gradient");
    if (n >= (1ll << 31)) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "too many points");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        (void)hipGetLastError();
        return vgi::fail(VG_ERR_NO_DEVICE, "no HIP device: CameraJacobian has no CPU fallback");
    }
    if (device < 0 || device >= n_dev) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "device index out of range");
    if (!n || (!dpdxi && !dfdxi)) return VG_OK;
    VG_HIP(hipSetDevice(device));
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    vg::CameraJacobianArgs a;
    vg::camera_jacobian_frame(T12, T23, a.L11, a.L12, a.L22);
    a.two = T23 ? 1 : 0;
    for (int k = 0; k < 10; k++) a.intr[k] = k < K ? intrinsics[k] : 0.;
    a.X2 = X2;
    a.grad = grad;
    a.dpdxi = dpdxi;
    a.dfdxi = dfdxi;
    a.n = (unsigned)n;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    switch (model) {
    case vg::kEUCM: hipLaunchKernelGGL((vg::vg_camera_jacobian_kernel<vg::kEUCM>), grid, blk, vgl::kLocalLds, st, a); break;
    case vg::kUCM: hipLaunchKernelGGL((vg::vg_camera_jacobian_kernel<vg::kUCM>), grid, blk, vgl::kLocalLds, st, a); break;
    default: hipLaunchKernelGGL((vg::vg_camera_jacobian_kernel<vg::kMEI>), grid, blk, vgl::kLocalLds, st, a); break;
    }
    VG_HIP(hipGetLastError());
    return VG_OK;
}

}  // extern "C"
