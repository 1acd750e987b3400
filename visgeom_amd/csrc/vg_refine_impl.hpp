// vg_refine_impl.hpp -- host side of vg_refine_poses (kernel: vg_pose_lm.hpp).  Included at the end of vg_capi.hip.
#pragma once

#include <mutex>

#include "vg_internal.hpp"
#include "vg_pose_lm.hpp"

namespace {

// ---- memory of the refinement calls, kept by the library between calls (vg_release_cached_memory) -----------------------------
// Round 5's wrapper paid eight hipMalloc + eight hipFree, five pageable uploads and four read-backs around a 0.25 ms kernel
// (call 1.0 ms).  Now: ONE device block [next | intr | board] and ONE pinned block [next | intr | board | poses | cost | it | term],
// grow-only, two events made once.  The small front is ONE upload; the per-image data -- a 6-vector read once at the start of an
// image's solve, 6 + 3 values written once at its end -- is read and written by the kernel IN the pinned block (mapped host
// memory): no copy in either direction, the results are there when the stream is idle.  For callers whose corners already live in
// HBM (a vg_problem's dataset, the calibration front end's CornerBlock) there is no observation traffic at all.
struct RefineScratch {
    int device = -1;
    char *dev = nullptr, *pin = nullptr, *pin_dev = nullptr;   // pin_dev: the pinned block as the device addresses it
    size_t cap = 0, dev_cap = 0;
    double *d_obs = nullptr;   // the host-pointer entry's observations (grow-only)
    size_t obs_cap = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // pinned ping-pong staging of vgi::upload_corners
    char *stage = nullptr;
    size_t stage_half = 0;
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    void drop()
    {
        if (dev) (void)hipFree(dev);
        if (pin) (void)hipHostFree(pin);
        if (d_obs) (void)hipFree(d_obs);
        if (stage) (void)hipHostFree(stage);
        for (hipEvent_t e : {e0, e1, stage_done[0], stage_done[1]})
            if (e) (void)hipEventDestroy(e);
        *this = RefineScratch();
    }
};
std::mutex g_refine_m;
RefineScratch g_refine;

int refine_scratch_for(int device, size_t bytes, size_t obs_bytes, size_t front_bytes = 0)
{
    RefineScratch &r = g_refine;
    if (r.device != device) r.drop();
    r.device = device;
    if (front_bytes > r.dev_cap) {   // the device block holds the front only: counter, intrinsics, board
        if (r.dev) (void)hipFree(r.dev);
        r.dev = nullptr;
        r.dev_cap = 0;
        const size_t want = front_bytes < 64 * 1024 ? (size_t)64 * 1024 : front_bytes * 2;
        VG_HIP(hipMalloc(&r.dev, want));
        r.dev_cap = want;
    }
    if (bytes > r.cap) {
        if (r.pin) (void)hipHostFree(r.pin);
        r.pin = r.pin_dev = nullptr;
        r.cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        VG_HIP(hipHostMalloc(&r.pin, want, hipHostMallocMapped));
        VG_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&r.pin_dev), r.pin, 0));
        r.cap = want;
    }
    if (obs_bytes > r.obs_cap) {
        if (r.d_obs) (void)hipFree(r.d_obs);
        r.d_obs = nullptr;
        r.obs_cap = 0;
        VG_HIP(hipMalloc(&r.d_obs, obs_bytes));
        r.obs_cap = obs_bytes;
    }
    if (!r.e0) {
        VG_HIP(hipEventCreate(&r.e0));
        VG_HIP(hipEventCreate(&r.e1));
    }
    return VG_OK;
}

inline size_t up256(size_t n) { return (n + 255) & ~(size_t)255; }

int check_refine_device(int device)
{
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return vgi::fail(VG_ERR_NO_DEVICE, "no HIP device available; visgeom_amd has no CPU fallback");
    if (device < 0 || device >= n_dev) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "device index out of range");
    return VG_OK;
}

}  // namespace

void vgi::refine_release_cached()
{
    std::lock_guard<std::mutex> lk(g_refine_m);
    g_refine.drop();
}

vgi::CornerBlock::~CornerBlock()
{
    if (d_obs) (void)hipFree(d_obs);
}

// The corners of a dataset into HBM, ONCE: `gather(first, count, dst)` writes images [first, first + count) as [image][N][2]
// into pinned staging (two halves in flight: the host's threads fill one while the other crosses the bus).
int vgi::upload_corners(int device, void *hip_stream, int64_t n_images, int n_points, const GatherFn &gather, std::shared_ptr<CornerBlock> *out)
{
    if (!out || n_images < 0 || n_points <= 0) return fail(VG_ERR_INVALID_ARGUMENT, "bad corner block");
    const int rc0 = check_refine_device(device);
    if (rc0 != VG_OK) return rc0;
    VG_HIP(hipSetDevice(device));
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    auto blk = std::make_shared<CornerBlock>();
    blk->device = device;
    blk->n_images = n_images;
    blk->N = n_points;
    const size_t per_image = sizeof(double) * 2 * (size_t)n_points;
    VG_HIP(hipMalloc(&blk->d_obs, per_image * (size_t)(n_images ? n_images : 1)));
    std::lock_guard<std::mutex> lk(g_refine_m);
    RefineScratch &r = g_refine;
    if (r.device != device) r.drop();
    r.device = device;
    const size_t half = (size_t)8 << 20;   // pinning costs ~80 us per MiB once: 16 MiB of staging, two chunks for 10 k images
    if (!r.stage) {
        VG_HIP(hipHostMalloc(&r.stage, 2 * half, hipHostMallocDefault));
        r.stage_half = half;
        VG_HIP(hipEventCreateWithFlags(&r.stage_done[0], hipEventDisableTiming));
        VG_HIP(hipEventCreateWithFlags(&r.stage_done[1], hipEventDisableTiming));
    }
    const int64_t per_chunk = (int64_t)(r.stage_half / per_image);
    if (per_chunk < 1) return fail(VG_ERR_INVALID_ARGUMENT, "a single image's corners exceed the staging buffer");
    // whichever way this is left, nothing of the library's staging may still be on the bus when the next caller fills it
    struct Drain {
        hipStream_t st;
        ~Drain() { (void)hipStreamSynchronize(st); }
    } drain{st};
    bool used[2] = {false, false};
    int k = 0;
    for (int64_t first = 0; first < n_images; first += per_chunk, k ^= 1) {
        const int64_t count = n_images - first < per_chunk ? n_images - first : per_chunk;
        if (used[k]) VG_HIP(hipEventSynchronize(r.stage_done[k]));
        double *dst = reinterpret_cast<double *>(r.stage + (size_t)k * r.stage_half);
        gather(first, count, dst);
        VG_HIP(hipMemcpyAsync(blk->d_obs + (size_t)first * 2 * n_points, dst, per_image * (size_t)count, hipMemcpyHostToDevice, st));
        VG_HIP(hipEventRecord(r.stage_done[k], st));
        used[k] = true;
    }
    VG_HIP(hipStreamSynchronize(st));
    *out = blk;
    return VG_OK;
}

// The refinement proper.  d_obs: [n_images][N][2] in HBM.  d_intr / d_board: device pointers, or NULL -> h_intr / h_board are
// uploaded with the poses.  poses / iterations / final_cost / termination: host arrays as in vg_refine_poses.
int vgi::refine_poses_resident(int device, void *hip_stream, int model, const double *d_intr, const double *h_intr, int n_points,
                               const double *d_board, const double *h_board, int64_t n_images, const double *d_obs, double *poses,
                               const vg_solve_options *options, int32_t *iterations, double *final_cost, int32_t *termination,
                               double *kernel_seconds, bool locked)
{
    if (kernel_seconds) *kernel_seconds = 0.;
    using vgi::fail;
    const int K = vg::num_intrinsics(model);
    if (K < 0) return fail(VG_ERR_INVALID_ARGUMENT, "unknown camera model");
    if ((!d_intr && !h_intr) || (!d_board && !h_board) || n_points <= 0 || n_images < 0 || (n_images > 0 && (!d_obs || !poses)))
        return fail(VG_ERR_INVALID_ARGUMENT, "NULL / empty argument");
    if (n_images > 0x3fffffff) return fail(VG_ERR_INVALID_ARGUMENT, "too many images for one launch");
    if (!n_images) return VG_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    vg_solve_options o;
    if (options) o = *options;
    else {
        vg_solve_options_init(&o);
        // what the reference's sub-problem runs with: Ceres' own defaults except max_num_iterations = 500 (:1148)
        o.max_num_iterations = 500;
        o.function_tolerance = 1e-6;
        o.gradient_tolerance = 1e-10;
        o.parameter_tolerance = 1e-8;
        o.soft_l1_scale = 25.;  // new SoftLOneLoss(25), :1143
    }
    std::unique_lock<std::mutex> lk(g_refine_m, std::defer_lock);
    if (!locked) lk.lock();
    const size_t n = (size_t)n_images, N = (size_t)n_points;
    // [next | intr | board | poses | cost | it | term]
    const size_t o_next = 0, o_intr = 256, o_board = o_intr + up256(sizeof(double) * K), o_poses = o_board + up256(sizeof(double) * 3 * N),
                 o_cost = o_poses + up256(sizeof(double) * 6 * n), o_it = o_cost + up256(sizeof(double) * n), o_term = o_it + up256(sizeof(int) * n),
                 total = o_term + up256(sizeof(int) * n);
    const int rcs = refine_scratch_for(device, total, 0, o_poses);
    if (rcs != VG_OK) return rcs;
    RefineScratch &r = g_refine;
    *reinterpret_cast<unsigned int *>(r.pin + o_next) = 0u;
    if (!d_intr) std::memcpy(r.pin + o_intr, h_intr, sizeof(double) * K);
    if (!d_board) std::memcpy(r.pin + o_board, h_board, sizeof(double) * 3 * N);
    std::memcpy(r.pin + o_poses, poses, sizeof(double) * 6 * n);
    // from here on work that reads and writes the library's own blocks is queued: no way out of this function without draining it
    struct Drain {
        hipStream_t st;
        bool armed;
        ~Drain()
        {
            if (armed) (void)hipStreamSynchronize(st);
        }
    } drain{st, true};
    VG_HIP(hipMemcpyAsync(r.dev, r.pin, o_poses, hipMemcpyHostToDevice, st));   // counter, intrinsics, board: ONE small copy
    vg::PoseLmArgs a;
    a.board = d_board ? d_board : reinterpret_cast<const double *>(r.dev + o_board);
    a.obs = d_obs;
    a.intr = d_intr ? d_intr : reinterpret_cast<const double *>(r.dev + o_intr);
    a.poses = reinterpret_cast<double *>(r.pin_dev + o_poses);   // mapped host memory: read once, written once per image
    a.iterations = iterations ? reinterpret_cast<int *>(r.pin_dev + o_it) : nullptr;
    a.final_cost = final_cost ? reinterpret_cast<double *>(r.pin_dev + o_cost) : nullptr;
    a.termination = termination ? reinterpret_cast<int *>(r.pin_dev + o_term) : nullptr;
    a.next = reinterpret_cast<unsigned int *>(r.dev + o_next);
    a.n_images = (unsigned int)n_images;
    a.N = (unsigned int)n_points;
    a.max_iter = o.max_num_iterations;
    a.a2 = o.soft_l1_scale > 0. ? o.soft_l1_scale * o.soft_l1_scale : 0.;
    a.ftol = o.function_tolerance;
    a.gtol = o.gradient_tolerance;
    a.ptol = o.parameter_tolerance;
    a.radius0 = o.initial_trust_region_radius;
    a.max_radius = o.max_trust_region_radius;
    a.min_radius = o.min_trust_region_radius;
    a.min_rel_decrease = o.min_relative_decrease;
    a.dmin = o.min_lm_diagonal;
    a.dmax = o.max_lm_diagonal;
    // a persistent grid: as many workgroups as are resident at once (two per CU at two waves per SIMD), never more than the images
    // need; every half-wave starts on the image of its position and takes further ones from the counter
    static int cu_count[64] = {0};   // per device, asked once (hipGetDeviceProperties is a millisecond)
    static int per_cu_of[64][3] = {{0}};
    if (device < 64 && !cu_count[device]) {
        int cus = 0;
        VG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        cu_count[device] = cus > 0 ? cus : 1;
    }
    const int n_cus = device < 64 ? cu_count[device] : 256;
    const int mi = model == VG_MODEL_EUCM ? 0 : model == VG_MODEL_UCM ? 1 : 2;
    int per_cu = device < 64 ? per_cu_of[device][mi] : 0;
    if (per_cu < 1) {
        const void *fn = model == VG_MODEL_EUCM ? reinterpret_cast<const void *>(vg::vg_pose_lm_kernel<vg::kEUCM>)
                         : model == VG_MODEL_UCM ? reinterpret_cast<const void *>(vg::vg_pose_lm_kernel<vg::kUCM>)
                                                 : reinterpret_cast<const void *>(vg::vg_pose_lm_kernel<vg::kMEI>);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, vg::kValuThreads, 0) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            per_cu = 1;
        }
        if (device < 64) per_cu_of[device][mi] = per_cu;
    }
    const int64_t wgs_needed = (n_images + vg::kValuImagesPerBlock - 1) / vg::kValuImagesPerBlock;
    const int64_t wgs_resident = (int64_t)per_cu * n_cus;
    const dim3 grid((unsigned int)(wgs_needed < wgs_resident ? wgs_needed : wgs_resident)), blk(vg::kValuThreads);
    if (kernel_seconds) VG_HIP(hipEventRecord(r.e0, st));
    switch (model) {
    case VG_MODEL_EUCM: hipLaunchKernelGGL(vg::vg_pose_lm_kernel<vg::kEUCM>, grid, blk, 0, st, a); break;
    case VG_MODEL_UCM: hipLaunchKernelGGL(vg::vg_pose_lm_kernel<vg::kUCM>, grid, blk, 0, st, a); break;
    default: hipLaunchKernelGGL(vg::vg_pose_lm_kernel<vg::kMEI>, grid, blk, 0, st, a); break;
    }
    VG_HIP(hipGetLastError());
    if (kernel_seconds) VG_HIP(hipEventRecord(r.e1, st));
    VG_HIP(hipStreamSynchronize(st));   // the kernel's stores to the mapped block are visible once the stream is idle
    drain.armed = false;
    std::memcpy(poses, r.pin + o_poses, sizeof(double) * 6 * n);
    if (iterations) std::memcpy(iterations, r.pin + o_it, sizeof(int) * n);
    if (final_cost) std::memcpy(final_cost, r.pin + o_cost, sizeof(double) * n);
    if (termination) std::memcpy(termination, r.pin + o_term, sizeof(int) * n);
    if (kernel_seconds) {
        float ms = 0.f;
        VG_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        *kernel_seconds = 1e-3 * (double)ms;
    }
    return VG_OK;
}

// vg_refine_poses: everything in host memory.  The corners are uploaded from the caller's own array into a buffer the library
// keeps (one copy; whether it is a staged pageable copy or a DMA from pinned memory is the caller's choice of allocation).
int vgi::refine_poses(int device, void *hip_stream, int model, const double *intrinsics, int n_points, const double *board, int64_t n_images,
                      const double *corners, double *poses, const vg_solve_options *options, int32_t *iterations, double *final_cost,
                      int32_t *termination, double *kernel_seconds)
{
    if (kernel_seconds) *kernel_seconds = 0.;
    if (vg::num_intrinsics(model) < 0) return fail(VG_ERR_INVALID_ARGUMENT, "unknown camera model");
    if (!intrinsics || !board || n_points <= 0 || n_images < 0 || (n_images > 0 && (!corners || !poses)))
        return fail(VG_ERR_INVALID_ARGUMENT, "NULL / empty argument");
    if (n_images > 0x3fffffff) return fail(VG_ERR_INVALID_ARGUMENT, "too many images for one launch");
    const int rc0 = check_refine_device(device);
    if (rc0 != VG_OK) return rc0;
    if (!n_images) return VG_OK;
    VG_HIP(hipSetDevice(device));
    std::lock_guard<std::mutex> lk(g_refine_m);
    const size_t obs_bytes = sizeof(double) * 2 * (size_t)n_points * (size_t)n_images;
    const int rcs = refine_scratch_for(device, 0, obs_bytes);
    if (rcs != VG_OK) return rcs;
    VG_HIP(hipMemcpyAsync(g_refine.d_obs, corners, obs_bytes, hipMemcpyHostToDevice, reinterpret_cast<hipStream_t>(hip_stream)));
    const int rc = refine_poses_resident(device, hip_stream, model, nullptr, intrinsics, n_points, nullptr, board, n_images, g_refine.d_obs, poses,
                                         options, iterations, final_cost, termination, kernel_seconds, /*locked=*/true);
    if (rc != VG_OK) (void)hipStreamSynchronize(reinterpret_cast<hipStream_t>(hip_stream));   // the corner upload reads the caller's array
    return rc;
}

// the poses of a dataset that is resident in a finalized problem: its observations, its board and the camera's CURRENT
// intrinsics are read where they lie in HBM; only the 6-vectors and the per-image results cross the bus
extern "C" int vg_dataset_refine_poses(vg_problem *p, int dataset_id, double *poses, const vg_solve_options *options, int32_t *iterations,
                                       double *final_cost, int32_t *termination, double *kernel_seconds)
{
    if (kernel_seconds) *kernel_seconds = 0.;
    const int rcv = vgi::valid_dataset(p, dataset_id);
    if (rcv != VG_OK) return rcv;
    if (!p->finalized) return vgi::fail(VG_ERR_STATE, "problem not finalized");
    const vgi::Dataset &d = p->dss[(size_t)dataset_id];
    if (d.n_blocks > 0 && !poses) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "poses is NULL");
    if (!d.n_blocks) return VG_OK;
    VG_HIP(hipSetDevice(p->device));
    const vgi::Camera &cam = p->cams[(size_t)d.camera];
    return vgi::refine_poses_resident(p->device, p->stream, cam.model, p->d_params + cam.offset, nullptr, d.N, d.d_board, nullptr, d.n_blocks,
                                      d.d_obs, poses, options, iterations, final_cost, termination, kernel_seconds, false);
}

extern "C" int vg_refine_poses(int device, void *hip_stream, int model, const double *intrinsics, int n_points, const double *board,
                               int64_t n_images, const double *corners, double *poses, const vg_solve_options *options,
                               int32_t *iterations, double *final_cost, int32_t *termination)
{
    return vgi::refine_poses(device, hip_stream, model, intrinsics, n_points, board, n_images, corners, poses, options, iterations,
                             final_cost, termination, nullptr);
}

extern "C" int vg_refine_poses_timed(int device, void *hip_stream, int model, const double *intrinsics, int n_points, const double *board,
                                     int64_t n_images, const double *corners, double *poses, const vg_solve_options *options,
                                     int32_t *iterations, double *final_cost, int32_t *termination, double *kernel_seconds)
{
    return vgi::refine_poses(device, hip_stream, model, intrinsics, n_points, board, n_images, corners, poses, options, iterations,
                             final_cost, termination, kernel_seconds);
}
