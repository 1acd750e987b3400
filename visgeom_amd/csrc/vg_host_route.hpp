// vg_host_route.hpp -- vg_dataset_evaluate_to_host: one evaluation delivered to HOST memory (the route of a Ceres
// EvaluationCallback, INTEGRATION.md section 2: ceres::Solve at src/calibration/unified_calibration.cpp:53 walks the residual
// blocks after ONE batched evaluation).  Included by vg_capi.hip (it uses that unit's launch helpers).
//
// The rows are 208 bytes per observation on a bus of ~55 GB/s: the evaluation is 36 us of kernel in front of 3.6 ms of copy at
// the headline size.  So the route is built around the copy:
//   * the dataset is cut into chunks of whole images (~32 MB of rows each); chunk k's emit launch runs on the problem's stream
//     into a device staging block laid out chunk-major ([residuals | jac_intr | jac_member ...] of the chunk, contiguous), an
//     event marks it ready and the copy stream takes it from there -- chunk k + 1 is evaluated while chunk k travels;
//   * destinations that are pinned (hipHostMalloc / hipHostRegister memory) receive their pieces straight from the copy engine;
//   * destinations that are NOT pinned -- the arrays Ceres allocates -- would be staged by the runtime in small synchronous
//     pieces (what a box measures then is anything from 7 to 20 GB/s): instead ONE copy per chunk lands in a pinned staging
//     block the library owns (allocated and touched at the first call), and the host's threads move the chunk into the
//     caller's arrays while the next chunk is on the bus.
// Chunks are small enough for plain stores (the rows wait in the Infinity Cache for the copy engine).
#pragma once

#include <cstring>

#include "vg_host_parallel.hpp"

namespace {

constexpr int64_t kHostChunkBytes = 32ll << 20;
constexpr int kHostMaxChunks = 256;

inline bool host_pointer_is_pinned(const void *ptr)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary host pointer: not an error of ours
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

}  // namespace

extern "C" int vg_dataset_evaluate_to_host(vg_problem *p, int dataset_id, double *residuals, double *jac_intr, double *const *jac_member)
{
    int rc = valid_dataset(p, dataset_id);
    if (rc != VG_OK) return rc;
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    Dataset &d = p->dss[dataset_id];
    d.epoch = (d.epoch + 1) & 0xFFFFFFull;
    if (d.epoch == 0) d.epoch = 1;
    if (!d.n_blocks) return VG_OK;
    if (!residuals) return fail(VG_ERR_INVALID_ARGUMENT, "residuals is NULL");
    VG_HIP(hipSetDevice(p->device));
    const Camera &cam = p->cams[d.camera];
    const int K = cam.K;
    bool want_jm[vg::kMaxChain] = {false};
    bool want_jac = jac_intr != nullptr;
    for (int l = 0; l < d.L; l++) {
        want_jm[l] = jac_member && jac_member[l];
        want_jac = want_jac || want_jm[l];
    }
    // doubles per image in a chunk: residuals, then the requested Jacobian arrays
    const size_t rows = 2 * (size_t)d.N;
    size_t per_image = rows;
    if (jac_intr) per_image += rows * K;
    for (int l = 0; l < d.L; l++)
        if (want_jm[l]) per_image += rows * 6;
    const long long chunk_hook = vgi::debug_hook(vgi::kHookHostChunkBytes);  // tests: many small chunks
    int64_t chunk_images = (chunk_hook > 0 ? (int64_t)chunk_hook : kHostChunkBytes) / (int64_t)(per_image * sizeof(double));
    if (chunk_images < 1) chunk_images = 1;
    if ((d.n_blocks + chunk_images - 1) / chunk_images > kHostMaxChunks) chunk_images = (d.n_blocks + kHostMaxChunks - 1) / kHostMaxChunks;
    {   // the launches index observations with 32 bits
        const int64_t max_blocks = (((int64_t)1 << 30) / d.N) > 0 ? ((int64_t)1 << 30) / d.N : 1;
        if (chunk_images > max_blocks) chunk_images = max_blocks;
    }
    const int n_chunks = (int)((d.n_blocks + chunk_images - 1) / chunk_images);

    // destinations: all pinned -> straight from the copy engine; otherwise through the library's pinned staging block.  (A few
    // megabytes are not worth a pinned block: the runtime's own staging moves them as fast as the set-up of ours would.)
    bool direct = host_pointer_is_pinned(residuals) && (!jac_intr || host_pointer_is_pinned(jac_intr));
    for (int l = 0; l < d.L && direct; l++)
        if (want_jm[l]) direct = host_pointer_is_pinned(jac_member[l]);
    if (!direct && !d.h_host_stage && chunk_hook <= 0 && per_image * (size_t)d.n_blocks * sizeof(double) < ((size_t)8 << 20)) direct = true;

    const size_t total = per_image * (size_t)d.n_blocks;
    if (d.d_host_stage_doubles < total) {
        if (d.d_host_stage) (void)hipFree(d.d_host_stage);
        d.d_host_stage = nullptr;
        d.d_host_stage_doubles = 0;
        VG_HIP(hipMalloc(&d.d_host_stage, sizeof(double) * total));
        d.d_host_stage_doubles = total;
    }
    if (!direct && d.h_host_stage_doubles < total) {
        if (d.h_host_stage) (void)hipHostFree(d.h_host_stage);
        d.h_host_stage = nullptr;
        d.h_host_stage_doubles = 0;
        VG_HIP(hipHostMalloc(&d.h_host_stage, sizeof(double) * total, hipHostMallocDefault));
        d.h_host_stage_doubles = total;
        // first touch now, by several threads, not inside the first timed copy
        vgpar::parallel_ranges(total, (size_t)1 << 20, [&](size_t b, size_t e, int) { std::memset(d.h_host_stage + b, 0, sizeof(double) * (e - b)); });
    }
    if (!d.host_copy_stream) VG_HIP(hipStreamCreateWithFlags(&d.host_copy_stream, hipStreamNonBlocking));
    while ((int)d.host_chunk_ready.size() < n_chunks) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        VG_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
        d.host_chunk_ready.push_back(e0);
        VG_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
        d.host_chunk_copied.push_back(e1);
    }

    const bool inline_chain = single_launch_dataset(p, d);
    if (!inline_chain && (rc = vgi::ensure_frames(p)) != VG_OK) return rc;
    hipStream_t cs = d.host_copy_stream;
    // From the first launch on, a failure must not return while copies into the caller's arrays (or the staging the next call
    // reuses) are still queued: whichever way the function is left, both streams are drained first.
    struct Drain {
        hipStream_t a, b;
        bool armed;
        ~Drain()
        {
            if (!armed) return;
            (void)hipStreamSynchronize(a);
            (void)hipStreamSynchronize(b);
        }
    } drain{p->stream, cs, true};
    // queue everything: evaluate chunk k, mark it, copy it behind the mark
    size_t off = 0;  // doubles into the staging blocks
    for (int k = 0; k < n_chunks; k++) {
        const int64_t b0 = (int64_t)k * chunk_images;
        const int64_t nb = d.n_blocks - b0 < chunk_images ? d.n_blocks - b0 : chunk_images;
        double *c_res = d.d_host_stage + off, *cur = c_res + rows * (size_t)nb;
        double *c_ji = nullptr, *c_jm[vg::kMaxChain] = {nullptr};
        if (jac_intr) {
            c_ji = cur;
            cur += rows * K * (size_t)nb;
        }
        for (int l = 0; l < d.L; l++)
            if (want_jm[l]) {
                c_jm[l] = cur;
                cur += rows * 6 * (size_t)nb;
            }
        vg::EmitArgs a;
        fill_emit_args_at(p, d, a, b0, nb, c_res, c_ji, c_jm);
        a.nt_stores = 0;
        switch (cam.model) {
        case VG_MODEL_EUCM: rc = launch_emit<vg::kEUCM>(p->stream, a, want_jac, inline_chain); break;
        case VG_MODEL_UCM: rc = launch_emit<vg::kUCM>(p->stream, a, want_jac, inline_chain); break;
        default: rc = launch_emit<vg::kMEI>(p->stream, a, want_jac, inline_chain); break;
        }
        if (rc != VG_OK) return rc;
        VG_HIP(hipEventRecord(d.host_chunk_ready[(size_t)k], p->stream));
        VG_HIP(hipStreamWaitEvent(cs, d.host_chunk_ready[(size_t)k], 0));
        const size_t chunk_doubles = per_image * (size_t)nb;
        if (direct) {
            VG_HIP(hipMemcpyAsync(residuals + rows * (size_t)b0, c_res, sizeof(double) * rows * (size_t)nb, hipMemcpyDeviceToHost, cs));
            if (jac_intr)
                VG_HIP(hipMemcpyAsync(jac_intr + rows * K * (size_t)b0, c_ji, sizeof(double) * rows * K * (size_t)nb, hipMemcpyDeviceToHost, cs));
            for (int l = 0; l < d.L; l++)
                if (want_jm[l])
                    VG_HIP(hipMemcpyAsync(jac_member[l] + rows * 6 * (size_t)b0, c_jm[l], sizeof(double) * rows * 6 * (size_t)nb,
                                          hipMemcpyDeviceToHost, cs));
        } else {
            VG_HIP(hipMemcpyAsync(d.h_host_stage + off, c_res, sizeof(double) * chunk_doubles, hipMemcpyDeviceToHost, cs));
            VG_HIP(hipEventRecord(d.host_chunk_copied[(size_t)k], cs));
        }
        off += chunk_doubles;
    }
    if (direct) {
        VG_HIP(hipStreamSynchronize(cs));
        drain.armed = false;   // the copy stream waited for every launch: nothing is in flight
        return VG_OK;
    }
    // the host's threads move chunk k into the caller's arrays while chunk k + 1 is on the bus
    off = 0;
    for (int k = 0; k < n_chunks; k++) {
        const int64_t b0 = (int64_t)k * chunk_images;
        const int64_t nb = d.n_blocks - b0 < chunk_images ? d.n_blocks - b0 : chunk_images;
        VG_HIP(hipEventSynchronize(d.host_chunk_copied[(size_t)k]));
        // the chunk as a list of (destination, source, doubles) pieces, cut for the threads by bytes
        struct Piece {
            double *dst;
            const double *src;
            size_t n;
        } pieces[2 + vg::kMaxChain];
        int np = 0;
        const double *src = d.h_host_stage + off;
        pieces[np++] = {residuals + rows * (size_t)b0, src, rows * (size_t)nb};
        src += rows * (size_t)nb;
        if (jac_intr) {
            pieces[np++] = {jac_intr + rows * K * (size_t)b0, src, rows * K * (size_t)nb};
            src += rows * K * (size_t)nb;
        }
        for (int l = 0; l < d.L; l++)
            if (want_jm[l]) {
                pieces[np++] = {jac_member[l] + rows * 6 * (size_t)b0, src, rows * 6 * (size_t)nb};
                src += rows * 6 * (size_t)nb;
            }
        const size_t chunk_doubles = per_image * (size_t)nb;
        vgpar::parallel_ranges(chunk_doubles, (size_t)1 << 18, [&](size_t b, size_t e, int) {
            size_t pos = 0;  // position of the piece's first double in the chunk
            for (int q = 0; q < np; q++) {
                const size_t lo = b > pos ? b : pos, hi = e < pos + pieces[q].n ? e : pos + pieces[q].n;
                if (lo < hi) std::memcpy(pieces[q].dst + (lo - pos), pieces[q].src + (lo - pos), sizeof(double) * (hi - lo));
                pos += pieces[q].n;
            }
        });
        off += chunk_doubles;
    }
    drain.armed = false;   // every chunk's copy event has been waited for
    return VG_OK;
}
