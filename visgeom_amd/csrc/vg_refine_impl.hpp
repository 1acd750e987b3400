// vg_refine_impl.hpp -- host side of vg_refine_poses (kernel: vg_pose_lm.hpp).  Included at the end of vg_capi.hip.
#pragma once

#include "vg_internal.hpp"
#include "vg_pose_lm.hpp"

int vgi::refine_poses(int device, void *hip_stream, int model, const double *intrinsics, int n_points, const double *board, int64_t n_images,
                      const double *corners, double *poses, const vg_solve_options *options, int32_t *iterations, double *final_cost,
                      int32_t *termination, double *kernel_seconds)
{
    if (kernel_seconds) *kernel_seconds = 0.;
    using vgi::fail;
    const int K = vg::num_intrinsics(model);
    if (K < 0) return fail(VG_ERR_INVALID_ARGUMENT, "unknown camera model");
    if (!intrinsics || !board || n_points <= 0 || n_images < 0 || (n_images > 0 && (!corners || !poses)))
        return fail(VG_ERR_INVALID_ARGUMENT, "NULL / empty argument");
    if (n_images > 0x3fffffff) return fail(VG_ERR_INVALID_ARGUMENT, "too many images for one launch");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(VG_ERR_NO_DEVICE, "no HIP device available; visgeom_amd has no CPU fallback");
    if (device < 0 || device >= n_dev) return fail(VG_ERR_INVALID_ARGUMENT, "device index out of range");
    if (!n_images) return VG_OK;
    VG_HIP(hipSetDevice(device));
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
    struct Bufs {
        double *board = nullptr, *obs = nullptr, *intr = nullptr, *poses = nullptr, *cost = nullptr;
        int *it = nullptr, *term = nullptr;
        unsigned int *next = nullptr;
        ~Bufs()
        {
            for (void *q : {(void *)board, (void *)obs, (void *)intr, (void *)poses, (void *)cost, (void *)it, (void *)term, (void *)next})
                if (q) (void)hipFree(q);
        }
    } d;
    const size_t n = (size_t)n_images, N = (size_t)n_points;
    VG_HIP(hipMalloc(&d.board, sizeof(double) * 3 * N));
    VG_HIP(hipMalloc(&d.obs, sizeof(double) * 2 * N * n));
    VG_HIP(hipMalloc(&d.intr, sizeof(double) * K));
    VG_HIP(hipMalloc(&d.poses, sizeof(double) * 6 * n));
    VG_HIP(hipMalloc(&d.cost, sizeof(double) * n));
    VG_HIP(hipMalloc(&d.it, sizeof(int) * n));
    VG_HIP(hipMalloc(&d.term, sizeof(int) * n));
    VG_HIP(hipMalloc(&d.next, sizeof(unsigned int)));
    VG_HIP(hipMemsetAsync(d.next, 0, sizeof(unsigned int), st));
    VG_HIP(hipMemcpyAsync(d.board, board, sizeof(double) * 3 * N, hipMemcpyHostToDevice, st));
    VG_HIP(hipMemcpyAsync(d.obs, corners, sizeof(double) * 2 * N * n, hipMemcpyHostToDevice, st));
    VG_HIP(hipMemcpyAsync(d.intr, intrinsics, sizeof(double) * K, hipMemcpyHostToDevice, st));
    VG_HIP(hipMemcpyAsync(d.poses, poses, sizeof(double) * 6 * n, hipMemcpyHostToDevice, st));
    vg::PoseLmArgs a;
    a.board = d.board;
    a.obs = d.obs;
    a.intr = d.intr;
    a.poses = d.poses;
    a.iterations = d.it;
    a.final_cost = d.cost;
    a.termination = d.term;
    a.next = d.next;
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
    if (device < 64 && !cu_count[device]) {
        int cus = 0;
        VG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        cu_count[device] = cus > 0 ? cus : 1;
    }
    const int n_cus = device < 64 ? cu_count[device] : 256;
    int per_cu = 0;
    const void *fn = model == VG_MODEL_EUCM ? reinterpret_cast<const void *>(vg::vg_pose_lm_kernel<vg::kEUCM>)
                     : model == VG_MODEL_UCM ? reinterpret_cast<const void *>(vg::vg_pose_lm_kernel<vg::kUCM>)
                                             : reinterpret_cast<const void *>(vg::vg_pose_lm_kernel<vg::kMEI>);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, vg::kValuThreads, 0) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    const int64_t wgs_needed = (n_images + vg::kValuImagesPerBlock - 1) / vg::kValuImagesPerBlock;
    const int64_t wgs_resident = (int64_t)per_cu * n_cus;
    const dim3 grid((unsigned int)(wgs_needed < wgs_resident ? wgs_needed : wgs_resident)), blk(vg::kValuThreads);
    struct Events {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events()
        {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } ev;
    if (kernel_seconds) {
        VG_HIP(hipEventCreate(&ev.e0));
        VG_HIP(hipEventCreate(&ev.e1));
        VG_HIP(hipEventRecord(ev.e0, st));
    }
    switch (model) {
    case VG_MODEL_EUCM: hipLaunchKernelGGL(vg::vg_pose_lm_kernel<vg::kEUCM>, grid, blk, 0, st, a); break;
    case VG_MODEL_UCM: hipLaunchKernelGGL(vg::vg_pose_lm_kernel<vg::kUCM>, grid, blk, 0, st, a); break;
    default: hipLaunchKernelGGL(vg::vg_pose_lm_kernel<vg::kMEI>, grid, blk, 0, st, a); break;
    }
    VG_HIP(hipGetLastError());
    if (kernel_seconds) VG_HIP(hipEventRecord(ev.e1, st));
    VG_HIP(hipMemcpyAsync(poses, d.poses, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, st));
    if (iterations) VG_HIP(hipMemcpyAsync(iterations, d.it, sizeof(int) * n, hipMemcpyDeviceToHost, st));
    if (final_cost) VG_HIP(hipMemcpyAsync(final_cost, d.cost, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    if (termination) VG_HIP(hipMemcpyAsync(termination, d.term, sizeof(int) * n, hipMemcpyDeviceToHost, st));
    VG_HIP(hipStreamSynchronize(st));
    if (kernel_seconds) {
        float ms = 0.f;
        VG_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
        *kernel_seconds = 1e-3 * (double)ms;
    }
    return VG_OK;
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
