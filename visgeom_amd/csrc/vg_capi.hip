// vg_capi.hip -- implementation of include/visgeom_amd.h (host side + kernel launches).
// Built with hipcc for gfx950 only.  No CPU fallback: every compute entry needs a HIP device.
#define VG_TU_CORE  // the non-template kernels this translation unit owns (the headers guard them by owner)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "vg_internal.hpp"
#include "vg_gram.hpp"
#include "vg_gram_valu.hpp"
#include "vg_transf_host.hpp"

namespace {

thread_local std::string g_err;

}  // namespace

int vgi::fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

using vgi::Camera;
using vgi::Dataset;
using vgi::Transform;
using vgi::fail;
using vgi::valid_dataset;

namespace {

int check_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(VG_ERR_NO_DEVICE, std::string("no HIP device available (") +
                                          (e == hipSuccess ? "device count 0" : hipGetErrorString(e)) +
                                          "); visgeom_amd has no CPU fallback");
    if (device < 0 || device >= n) return fail(VG_ERR_INVALID_ARGUMENT, "device index out of range");
    return VG_OK;
}

void free_dataset(Dataset &d)
{
    if (d.d_board) (void)hipFree(d.d_board);
    if (d.d_obs && !d.resident) (void)hipFree(d.d_obs);   // a resident block is freed by its last owner
    d.resident.reset();
    if (d.d_frames) (void)hipFree(d.d_frames);
    if (d.d_seq) (void)hipFree(d.d_seq);
    if (d.d_failed) (void)hipFree(d.d_failed);
    if (d.d_partials) (void)hipFree(d.d_partials);
    if (d.d_wg_partials) (void)hipFree(d.d_wg_partials);
    d.d_partials = d.d_wg_partials = nullptr;
    if (d.d_host_stage) (void)hipFree(d.d_host_stage);
    if (d.h_host_stage) (void)hipHostFree(d.h_host_stage);
    d.d_host_stage = d.h_host_stage = nullptr;
    d.d_host_stage_doubles = d.h_host_stage_doubles = 0;
    for (hipEvent_t e : d.host_chunk_ready) (void)hipEventDestroy(e);
    for (hipEvent_t e : d.host_chunk_copied) (void)hipEventDestroy(e);
    d.host_chunk_ready.clear();
    d.host_chunk_copied.clear();
    if (d.host_copy_stream) (void)hipStreamDestroy(d.host_copy_stream);
    d.host_copy_stream = nullptr;
    d.d_board = d.d_obs = d.d_frames = nullptr;
    d.d_seq = nullptr;
    d.d_failed = nullptr;
}

template <int MODEL>
size_t emit_lds_bytes(bool want_jac, bool frames_lds, int N, int frame_stride)
{
    size_t bytes = 0;
    // the tile region is always reserved so the frame region's offset does not depend on WANT_JAC
    bytes += (size_t)(vg::kEmitThreads / vg::kWave) * vg::emit_stage_doubles_per_wave<MODEL>() * sizeof(double);
    (void)want_jac;
    if (frames_lds) bytes += (size_t)(vg::kEmitThreads / N + 2) * frame_stride * sizeof(double);
    return bytes;
}

// Largest evaluation (bytes of residuals + Jacobian rows + observations per launch) for which the emit kernel walks a
// single-member chain itself.  The in-kernel walk saves the chain-prep launch (~7 us + its boundary) and the frames' round trip
// through memory; it costs every workgroup ~2 us in front of its first store.  While the launch is absorbed by the Infinity Cache
// that is hidden -- whole step, same box, alternating (profiles/r06n_inline_vs_prep.txt, in-kernel walk / prep + emit): EUCM
// 35 k images 110 / 121 us, 50 k 156 / 167, 75 k (1.6 GB) 231 / 241; Mei 40 k 151 / 159, 60 k 248 / 257 -- but once the launch streams
// to DRAM the stores are latency bound and a workgroup that waits two microseconds before storing is bytes missing in flight:
// 85 k images (1.8 GB) 301 / 279 us, 100 k 430 / 360.  With the short walk in the tile (profiles/r06s_inline_vs_prep_fastwalk.txt, another
// box): 75 k 212 / 258, 85 k 239 / 304, 100 k 372 / 368, 150 k 556 / 549; Mei 70 k (1.9 GB) 306 / 315, 80 k 350 / 384 -- where the cache-assisted
// range ends depends on the box; 2.0 GB sits between the two.  Rounds 3-5 (before non-temporal stores, then before these A/Bs): 288 MB, 600 MB.
int64_t inline_chain_max_bytes()
{
    const long long h = vgi::debug_hook(vgi::kHookInlineChainMaxBytes);
    return h ? (int64_t)h : (int64_t)2000000000;
}

// Smallest output of a launch (bytes of residuals + Jacobian rows) that is written with non-temporal stores: everything that
// does not fit the 256 MiB Infinity Cache next to the observations it reads (stream_store16 in vg_kernels.hpp).
int64_t emit_nt_min_bytes()
{
    const long long h = vgi::debug_hook(vgi::kHookEmitNtMinBytes);
    return h ? (int64_t)h : (int64_t)230000000;
}

// Tile map of an emit launch by its output: one contiguous eighth per XCD while the launch stays inside or near the Infinity Cache
// (<= 1.2 GB: same box, alternating, the eighths are level with the windows for EUCM and 2 % ahead for Mei at 10 k images,
// profiles/r06l_headline_map_ab.txt), windows of 8 x kEmitMapWindow tiles beyond, where the eighths fall into their slow mode on
// most boxes (from ~1.6 GB; profiles/r06_emit_drop.md).  hook emit_map_window: W > 0 that window, -1 the eighths, whatever the size.
unsigned int emit_map_window(int64_t launch_output_bytes)
{
    const long long mw = vgi::debug_hook(vgi::kHookEmitMapWindow);
    if (mw) return mw > 0 ? (unsigned int)mw : 0u;
    return launch_output_bytes >= (int64_t)1200000000 ? vg::kEmitMapWindow : 0u;
}

int64_t emit_output_bytes(const vg::EmitArgs &a, int K)
{
    int64_t per_obs = 16;
    if (a.jac_intr) per_obs += 16 * K;
    for (int l = 0; l < a.L; l++)
        if (a.jac_member[l]) per_obs += 96;
    return per_obs * (int64_t)a.n_obs;
}

bool emit_frames_in_lds(int N, int frame_stride)
{
    const int max_frames = vg::kEmitThreads / N + 2;
    return (size_t)max_frames * frame_stride * sizeof(double) <= 32 * 1024;
}

template <int MODEL>
int launch_emit(hipStream_t stream, const vg::EmitArgs &a, bool want_jac, bool inline_chain)
{
    const bool frames_lds = emit_frames_in_lds((int)a.N, a.frame_stride_d);
    const size_t lds = emit_lds_bytes<MODEL>(want_jac, frames_lds, (int)a.N, a.frame_stride_d);
    const unsigned int grid = (a.n_obs + vg::kEmitThreads - 1) / vg::kEmitThreads;
    if (inline_chain) {  // the caller checked: one DIRECT member, frames fit the LDS
        if (want_jac) hipLaunchKernelGGL((vg::vg_emit_kernel<MODEL, true, true, true>), dim3(grid), dim3(vg::kEmitThreads), lds, stream, a);
        else hipLaunchKernelGGL((vg::vg_emit_kernel<MODEL, false, true, true>), dim3(grid), dim3(vg::kEmitThreads), lds, stream, a);
        VG_HIP(hipGetLastError());
        return VG_OK;
    }
    if (want_jac) {
        if (frames_lds)
            hipLaunchKernelGGL((vg::vg_emit_kernel<MODEL, true, true>), dim3(grid), dim3(vg::kEmitThreads), lds, stream, a);
        else
            hipLaunchKernelGGL((vg::vg_emit_kernel<MODEL, true, false>), dim3(grid), dim3(vg::kEmitThreads), lds, stream, a);
    } else {
        if (frames_lds)
            hipLaunchKernelGGL((vg::vg_emit_kernel<MODEL, false, true>), dim3(grid), dim3(vg::kEmitThreads), lds, stream, a);
        else
            hipLaunchKernelGGL((vg::vg_emit_kernel<MODEL, false, false>), dim3(grid), dim3(vg::kEmitThreads), lds, stream, a);
    }
    VG_HIP(hipGetLastError());
    return VG_OK;
}

bool dataset_can_inline_chain(const vg_problem *p, const Dataset &d)
{
    return d.L == 1 && d.status[0] == VG_TRANSFORM_DIRECT && emit_frames_in_lds(d.N, d.frame_stride) &&
           d.n_blocks * (int64_t)d.N * (32 + 16 * (p->cams[d.camera].K + 6)) <= inline_chain_max_bytes();
}

// The route of a dataset is a property of THAT dataset (and of the test hook), never of its neighbours or of what ran
// before: a block evaluated on its own, inside a block group or inside a rig problem gets the same bits.
bool single_launch_dataset(const vg_problem *p, const Dataset &d)
{
    return !p->force_prepared_frames && dataset_can_inline_chain(p, d);
}

// blocks [b0, b0 + nb) of a dataset; the output pointers are those of block b0 (chunk-local)
void fill_emit_args_at(const vg_problem *p, const Dataset &d, vg::EmitArgs &a, int64_t b0, int64_t nb, double *res_b0, double *ji_b0,
                       double *const *jm_b0)
{
    const Camera &cam = p->cams[d.camera];
    a.frames = d.d_frames + (size_t)b0 * d.frame_stride;
    a.board = d.d_board;
    a.obs = d.d_obs + (size_t)b0 * 2 * d.N;
    a.intr = p->d_params + cam.offset;
    a.res = res_b0;
    a.jac_intr = ji_b0;
    for (int l = 0; l < vg::kMaxChain; l++) a.jac_member[l] = (l < d.L && jm_b0) ? jm_b0[l] : nullptr;
    a.failed = d.d_failed;
    a.epoch = d.epoch;
    a.n_obs = (unsigned int)(nb * d.N);
    a.N = (unsigned int)d.N;
    a.L = d.L;
    a.frame_stride_d = d.frame_stride;
    a.chain_params = d.L ? p->d_params + d.chain.base[0] : nullptr;
    a.chain_stride = d.L ? d.chain.stride[0] : 0;
    a.seq_index = d.seq_identity ? nullptr : d.d_seq + b0;
    a.first_block = b0;
    a.nt_stores = emit_output_bytes(a, cam.K) >= emit_nt_min_bytes() ? 1 : 0;  // a merged launch decides for all its datasets together
    a.map_window = emit_map_window(emit_output_bytes(a, cam.K));   // likewise
}

// the same with whole-dataset arrays: block b0's rows lie b0 blocks into each of them
void fill_emit_args(const vg_problem *p, const Dataset &d, vg::EmitArgs &a, int64_t b0, int64_t nb, double *residuals,
                    double *jac_intr, double *const *jac_member)
{
    const int K = p->cams[d.camera].K;
    double *jm[vg::kMaxChain] = {nullptr};
    for (int l = 0; l < d.L; l++)
        if (jac_member && jac_member[l]) jm[l] = jac_member[l] + (size_t)b0 * 2 * d.N * 6;
    fill_emit_args_at(p, d, a, b0, nb, residuals + (size_t)b0 * 2 * d.N, jac_intr ? jac_intr + (size_t)b0 * 2 * d.N * K : nullptr, jm);
}

}  // namespace


#ifdef VG_DEBUG_HOOKS
namespace {
long long g_debug_hooks[vgi::kHookCount] = {0};
const char *const kDebugHookNames[vgi::kHookCount] = {"inline_chain_max_bytes", "gram_force_mfma", "gram_ch1", "gram_no_merge", "max_obs_per_launch",
                                                      "solver_timing", "solver_host_loop", "solver_device_loop", "solver_no_speculation", "emit_equal_tiles", "schur_private_gather", "solver_event_wait", "solver_no_fold_frames", "solver_one_wave_fold", "solver_fold_max_groups", "emit_nt_min_bytes", "host_chunk_bytes", "gram_stamps", "gram_persistent", "emit_map_window"};
}  // namespace
long long vgi::debug_hook(vgi::DebugHook h) { return g_debug_hooks[h]; }
#endif

extern "C" {

#ifdef VG_DEBUG_HOOKS   // the production library does not export the entry at all (tests/test_capi_cpu.py)
int vg_debug_set(const char *name, long long value)
{
    if (!name) return fail(VG_ERR_INVALID_ARGUMENT, "name is NULL");
    for (int k = 0; k < vgi::kHookCount; k++)
        if (std::strcmp(name, kDebugHookNames[k]) == 0) {
            g_debug_hooks[k] = value;
            return VG_OK;
        }
    return fail(VG_ERR_INVALID_ARGUMENT, std::string("unknown debug hook: ") + name);
}
#endif

int vg_abi_version(void) { return VG_ABI_VERSION; }

const char *vg_last_error(void) { return g_err.c_str(); }

int vg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int vg_num_intrinsics(int model) { return vg::num_intrinsics(model); }

int vg_intrinsic_bounds(int model, int idx, double *lower, double *upper)
{
    const int K = vg::num_intrinsics(model);
    if (K < 0 || idx < 0 || idx >= K || !lower || !upper) return fail(VG_ERR_INVALID_ARGUMENT, "bad model / index");
    double lo = 1., hi = 1e5;  // "the rest": focal lengths and centre
    if (model == VG_MODEL_EUCM) {          // eucm.h:228-246
        if (idx == 0) { lo = 0.; hi = 1.; }
        else if (idx == 1) { lo = 0.1; hi = 10.; }
    } else if (model == VG_MODEL_UCM) {    // ucm.h:199-215
        if (idx == 0) { lo = 0.; hi = 3.; }
    } else {                               // mei.h:287-313
        if (idx == 0) { lo = 0.; hi = 3.; }
        else if (idx <= 5) { lo = -10.; hi = 10.; }
    }
    *lower = lo;
    *upper = hi;
    return VG_OK;
}

/* ------------------------------------------------------------------------------------------ problem */

int vg_problem_create(vg_problem **out, int device, void *hip_stream)
{
    if (!out) return fail(VG_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int rc = check_device(device);
    if (rc != VG_OK) return rc;
    VG_HIP(hipSetDevice(device));
    vg_problem *p = new (std::nothrow) vg_problem();
    if (!p) return fail(VG_ERR_ALLOC, "out of host memory");
    p->device = device;
    p->stream = reinterpret_cast<hipStream_t>(hip_stream);
    *out = p;
    return VG_OK;
}

void vg_problem_destroy(vg_problem *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (auto &d : p->dss) free_dataset(d);
    if (p->d_prep) (void)hipFree(p->d_prep);
    if (p->d_params) (void)hipFree(p->d_params);
    delete p;
}

int vg_problem_add_camera(vg_problem *p, int model, const double *intrinsics, int constant, int *camera_id)
{
    if (!p || !intrinsics) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (p->finalized) return fail(VG_ERR_STATE, "problem already finalized");
    const int K = vg::num_intrinsics(model);
    if (K < 0) return fail(VG_ERR_INVALID_ARGUMENT, "unknown camera model");  // :177 throws
    Camera c;
    c.model = model;
    c.K = K;
    c.constant = constant != 0;
    c.init.assign(intrinsics, intrinsics + K);
    p->cams.push_back(c);
    if (camera_id) *camera_id = (int)p->cams.size() - 1;
    return VG_OK;
}

int vg_problem_add_transform(vg_problem *p, int is_global, int constant, int count, const double *values,
                             int *transform_id)
{
    if (!p) return fail(VG_ERR_INVALID_ARGUMENT, "problem is NULL");
    if (p->finalized) return fail(VG_ERR_STATE, "problem already finalized");
    if (is_global) count = 1;
    if (count < 0) return fail(VG_ERR_INVALID_ARGUMENT, "negative transform count");
    Transform t;
    t.global = is_global != 0;
    t.constant = constant != 0;
    t.count = count;
    t.init.assign((size_t)count * 6, 0.);
    if (values) std::memcpy(t.init.data(), values, sizeof(double) * 6 * (size_t)count);
    p->tfs.push_back(t);
    if (transform_id) *transform_id = (int)p->tfs.size() - 1;
    return VG_OK;
}

}  // extern "C"

namespace {
int add_dataset_common(vg_problem *p, int camera_id, int chain_len, const int *transform_ids, const int *status,
                       int n_points, const double *board, int64_t n_images, const int32_t *image_index,
                       const double *corners, const std::shared_ptr<vgi::CornerBlock> &resident, int *dataset_id)
{
    if (!p) return fail(VG_ERR_INVALID_ARGUMENT, "problem is NULL");
    if (p->finalized) return fail(VG_ERR_STATE, "problem already finalized");
    if (camera_id < 0 || camera_id >= (int)p->cams.size()) return fail(VG_ERR_INVALID_ARGUMENT, "camera id out of range");
    if (chain_len < 0 || chain_len > VG_MAX_CHAIN)
        return fail(VG_ERR_INVALID_ARGUMENT, "chain length must be in [0, 5]");  // :566-567 throws above 5
    if (chain_len > 0 && (!transform_ids || !status)) return fail(VG_ERR_INVALID_ARGUMENT, "chain arrays are NULL");
    if (n_points <= 0 || !board) return fail(VG_ERR_INVALID_ARGUMENT, "empty board");
    if (n_images < 0) return fail(VG_ERR_INVALID_ARGUMENT, "negative image count");   // corners NULL: zero observations (see the header)
    Dataset d;
    d.camera = camera_id;
    d.L = chain_len;
    d.N = n_points;
    d.n_blocks = n_images;
    int64_t min_seq = -1;
    for (int l = 0; l < chain_len; l++) {
        const int t = transform_ids[l];
        if (t < 0 || t >= (int)p->tfs.size()) return fail(VG_ERR_INVALID_ARGUMENT, "transform id out of range");
        if (status[l] != VG_TRANSFORM_DIRECT && status[l] != VG_TRANSFORM_INVERSE)
            return fail(VG_ERR_INVALID_ARGUMENT, "status must be DIRECT or INVERSE");
        d.tids[l] = t;
        d.status[l] = status[l];
        if (!p->tfs[t].global && (min_seq < 0 || p->tfs[t].count < min_seq)) min_seq = p->tfs[t].count;
    }
    d.h_seq.resize((size_t)n_images);
    for (int64_t i = 0; i < n_images; i++) {
        const int64_t idx = image_index ? image_index[i] : i;
        if (idx < 0 || (min_seq >= 0 && idx >= min_seq))
            return fail(VG_ERR_INVALID_ARGUMENT, "image index outside the sequence transform");
        d.h_seq[(size_t)i] = (int32_t)idx;
    }
    d.h_board.assign(board, board + 3 * (size_t)n_points);
    if (resident) {
        if (resident->device != p->device || resident->N != n_points || resident->n_images < n_images)
            return fail(VG_ERR_INVALID_ARGUMENT, "the resident corner block does not fit the dataset (device, board size or image count)");
        d.resident = resident;
    } else {
        d.zero_obs = corners == nullptr;
        if (corners) d.h_obs.assign(corners, corners + (size_t)n_images * 2 * n_points);
    }
    p->dss.push_back(std::move(d));
    if (dataset_id) *dataset_id = (int)p->dss.size() - 1;
    return VG_OK;
}
}  // namespace

// a dataset whose observations are zeros (cleared on the device, nothing uploaded): its residuals are the projections
// themselves, which is how writeImageResidual (unified_calibration.cpp:1186-1292) projects the board
int vgi::problem_add_projection_dataset(vg_problem *p, int camera_id, int chain_len, const int *transform_ids, const int *status, int n_points,
                                        const double *board, int64_t n_images, const int32_t *image_index, int *dataset_id)
{
    return add_dataset_common(p, camera_id, chain_len, transform_ids, status, n_points, board, n_images, image_index, nullptr, nullptr, dataset_id);
}

int vgi::problem_add_dataset_resident(vg_problem *p, int camera_id, int chain_len, const int *transform_ids, const int *status, int n_points,
                                      const double *board, int64_t n_images, const int32_t *image_index,
                                      const std::shared_ptr<CornerBlock> &corners, int *dataset_id)
{
    if (!corners) return fail(VG_ERR_INVALID_ARGUMENT, "corner block is NULL");
    return add_dataset_common(p, camera_id, chain_len, transform_ids, status, n_points, board, n_images, image_index, nullptr, corners, dataset_id);
}

extern "C" {

int vg_problem_add_dataset(vg_problem *p, int camera_id, int chain_len, const int *transform_ids, const int *status,
                           int n_points, const double *board, int64_t n_images, const int32_t *image_index,
                           const double *corners, int *dataset_id)
{
    if (n_images > 0 && !corners) return fail(VG_ERR_INVALID_ARGUMENT, "corners is NULL");
    return add_dataset_common(p, camera_id, chain_len, transform_ids, status, n_points, board, n_images, image_index, corners, nullptr, dataset_id);
}

int vg_problem_add_transformation_prior(vg_problem *p, int transform_id, const double *stiffness)
{
    if (!p || !stiffness) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (p->finalized) return fail(VG_ERR_STATE, "problem already finalized");
    if (transform_id < 0 || transform_id >= (int)p->tfs.size()) return fail(VG_ERR_INVALID_ARGUMENT, "transform id out of range");
    const Transform &t = p->tfs[transform_id];
    // a sequence transform gets the block on its element 0: getTransformData(name) defaults to index 0 (:826)
    if (!t.global && t.count < 1) return fail(VG_ERR_INVALID_ARGUMENT, "the sequence transform is empty");
    vgi::Prior pr;
    pr.tf = transform_id;
    for (int k = 0; k < 6; k++) pr.xi[k] = t.init[k];
    const vg::RotTrig g = vg::rot_trig(pr.xi + 3, true, true);
    vg::rotation_matrix(pr.xi + 3, 1., g, pr.R);   // _R(_xiPrior.rotMat())
    double M[9];
    vg::inter_omega_rot(pr.xi + 3, g, M);          // interOmegaRot(_xiPrior.rot())
    for (int k = 0; k < 36; k++) pr.A[k] = 0.;
    for (int k = 0; k < 3; k++) pr.A[6 * k + k] = stiffness[k];
    for (int r = 0; r < 3; r++)                    // bottomRightCorner = diag(stiffness[3..5]) * M
        for (int c = 0; c < 3; c++) pr.A[6 * (3 + r) + 3 + c] = stiffness[3 + r] * M[3 * r + c];
    p->priors.push_back(pr);
    return VG_OK;
}

int vg_problem_add_odometry_prior(vg_problem *p, int transform_id, int64_t index, double err_v, double err_w, double lambda,
                                  const double *xi1, const double *xi2)
{
    if (!p || !xi1 || !xi2) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (p->finalized) return fail(VG_ERR_STATE, "problem already finalized");
    if (transform_id < 0 || transform_id >= (int)p->tfs.size()) return fail(VG_ERR_INVALID_ARGUMENT, "transform id out of range");
    const Transform &t = p->tfs[transform_id];
    if (t.global) return fail(VG_ERR_INVALID_ARGUMENT, "Odometry must be a sequence");  // :749-752
    if (index < 0 || index + 1 >= t.count) return fail(VG_ERR_INVALID_ARGUMENT, "odometry index outside the sequence");
    if (!(lambda > 0.)) return fail(VG_ERR_INVALID_ARGUMENT, "lambda must be positive");
    p->odoms.push_back(vgodo::make_block(transform_id, index, err_v, err_w, lambda, xi1, xi2));
    return VG_OK;
}

int vg_problem_add_parameter_block(vg_problem *p, int size, const double *values, int constant, int *block_id)
{
    if (!p || !values) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (p->finalized) return fail(VG_ERR_STATE, "problem already finalized");
    if (size < 1 || size > 16) return fail(VG_ERR_INVALID_ARGUMENT, "parameter block size must be in [1, 16]");
    vgi::ParamBlock b;
    b.size = size;
    b.constant = constant != 0;
    b.init.assign(values, values + size);
    p->pblocks.push_back(b);
    if (block_id) *block_id = (int)p->pblocks.size() - 1;
    return VG_OK;
}

int64_t vg_problem_parameter_block_offset(const vg_problem *p, int block_id)
{
    if (!p || !p->finalized || block_id < 0 || block_id >= (int)p->pblocks.size()) return -1;
    return p->pblocks[block_id].offset;
}

int vg_problem_add_odometry_cost(vg_problem *p, int transform_id, int64_t index, double err_v, double err_w, double lambda,
                                 int n_steps, const double *delta_q, int param_block_id)
{
    if (!p || !delta_q) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (p->finalized) return fail(VG_ERR_STATE, "problem already finalized");
    if (transform_id < 0 || transform_id >= (int)p->tfs.size()) return fail(VG_ERR_INVALID_ARGUMENT, "transform id out of range");
    const Transform &t = p->tfs[transform_id];
    if (t.global) return fail(VG_ERR_INVALID_ARGUMENT, "Odometry must be a sequence");  // :667-670
    if (index < 0 || index + 1 >= t.count) return fail(VG_ERR_INVALID_ARGUMENT, "odometry index outside the sequence");
    if (!(lambda > 0.)) return fail(VG_ERR_INVALID_ARGUMENT, "lambda must be positive");
    if (n_steps < 1) return fail(VG_ERR_INVALID_ARGUMENT, "an odometry interval needs at least one wheel increment");
    if (param_block_id < 0 || param_block_id >= (int)p->pblocks.size() || p->pblocks[param_block_id].size != 3)
        return fail(VG_ERR_INVALID_ARGUMENT, "the odometry intrinsics must be a parameter block of size 3");
    p->odoms.push_back(vgodo::make_cost_block(transform_id, index, err_v, err_w, lambda, delta_q, n_steps,
                                              p->pblocks[param_block_id].init.data(), param_block_id));
    return VG_OK;
}

int vg_odometry_cost_evaluate(double err_v, double err_w, double lambda, int n_steps, const double *delta_q, const double *intr_prior,
                              const double *xi1, const double *xi2, const double *intr, double *zeta_prior, double *residual,
                              double *J1, double *J2, double *J3)
{
    if (!delta_q || !intr_prior || !xi1 || !xi2 || !intr || !residual) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!(lambda > 0.) || n_steps < 1) return fail(VG_ERR_INVALID_ARGUMENT, "lambda must be positive, n_steps >= 1");
    const vgodo::Block b = vgodo::make_cost_block(0, 0, err_v, err_w, lambda, delta_q, n_steps, intr_prior, 0);
    if (zeta_prior) std::memcpy(zeta_prior, b.zeta, sizeof b.zeta);
    vgodo::evaluate_cost(b, xi1, xi2, intr, residual, J1, J2, J3);
    return VG_OK;
}

int vg_odometry_prior_evaluate(double err_v, double err_w, double lambda, const double *xi1_odom, const double *xi2_odom,
                               const double *xi1, const double *xi2, double *residual, double *J1, double *J2)
{
    if (!xi1_odom || !xi2_odom || !xi1 || !xi2 || !residual) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!(lambda > 0.)) return fail(VG_ERR_INVALID_ARGUMENT, "lambda must be positive");
    const vgodo::Block b = vgodo::make_block(0, 0, err_v, err_w, lambda, xi1_odom, xi2_odom);
    vgodo::evaluate(b, xi1, xi2, residual, J1, J2);
    return VG_OK;
}

int vg_problem_set_pose_constant(vg_problem *p, int transform_id, int64_t index)
{
    if (!p) return fail(VG_ERR_INVALID_ARGUMENT, "problem is NULL");
    if (p->finalized) return fail(VG_ERR_STATE, "problem already finalized");
    if (transform_id < 0 || transform_id >= (int)p->tfs.size()) return fail(VG_ERR_INVALID_ARGUMENT, "transform id out of range");
    const Transform &t = p->tfs[transform_id];
    if (t.global || index < 0 || index >= t.count) return fail(VG_ERR_INVALID_ARGUMENT, "not an element of a sequence transform");
    p->const_poses.emplace_back(transform_id, index);
    return VG_OK;
}

int vg_problem_finalize(vg_problem *p)
{
    if (!p) return fail(VG_ERR_INVALID_ARGUMENT, "problem is NULL");
    if (p->finalized) return fail(VG_ERR_STATE, "problem already finalized");
    VG_HIP(hipSetDevice(p->device));
    int64_t off = 0;
    for (auto &c : p->cams) { c.offset = off; off += c.K; }
    for (auto &t : p->tfs) { t.offset = off; off += 6 * t.count; }
    for (auto &b : p->pblocks) { b.offset = off; off += b.size; }
    p->n_params = off;
    std::vector<double> h((size_t)off, 0.);
    for (auto &c : p->cams) std::memcpy(h.data() + c.offset, c.init.data(), sizeof(double) * c.K);
    for (auto &t : p->tfs)
        if (t.count) std::memcpy(h.data() + t.offset, t.init.data(), sizeof(double) * 6 * (size_t)t.count);
    for (auto &b : p->pblocks) std::memcpy(h.data() + b.offset, b.init.data(), sizeof(double) * (size_t)b.size);
    VG_HIP(hipMalloc(&p->d_params, sizeof(double) * (size_t)(off > 0 ? off : 1)));
    if (off) VG_HIP(hipMemcpy(p->d_params, h.data(), sizeof(double) * (size_t)off, hipMemcpyHostToDevice));

    for (auto &d : p->dss) {
        d.frame_stride = vg::frame_stride(d.L);
        d.chain.L = d.L;
        for (int l = 0; l < vg::kMaxChain; l++) {
            d.chain.status[l] = 0;
            d.chain.base[l] = 0;
            d.chain.stride[l] = 0;
        }
        for (int l = 0; l < d.L; l++) {
            const Transform &t = p->tfs[d.tids[l]];
            d.chain.status[l] = d.status[l];
            d.chain.base[l] = t.offset;
            d.chain.stride[l] = t.global ? 0 : 6;
        }
        const size_t nb = (size_t)d.n_blocks;
        VG_HIP(hipMalloc(&d.d_board, sizeof(double) * 3 * (size_t)d.N));
        VG_HIP(hipMemcpy(d.d_board, d.h_board.data(), sizeof(double) * 3 * (size_t)d.N, hipMemcpyHostToDevice));
        if (d.resident) d.d_obs = d.resident->d_obs;
        else VG_HIP(hipMalloc(&d.d_obs, sizeof(double) * (nb ? nb : 1) * 2 * d.N));
        VG_HIP(hipMalloc(&d.d_seq, sizeof(int32_t) * (nb ? nb : 1)));
        VG_HIP(hipMalloc(&d.d_frames, sizeof(double) * (nb ? nb : 1) * d.frame_stride));
        VG_HIP(hipMalloc(&d.d_failed, sizeof(unsigned long long)));
        VG_HIP(hipMemset(d.d_failed, 0, sizeof(unsigned long long)));
        if (nb) {
            if (d.resident) {}   // already in HBM (vgi::upload_corners)
            else if (d.zero_obs) VG_HIP(hipMemset(d.d_obs, 0, sizeof(double) * nb * 2 * d.N));   // a projection dataset: r = proj - 0
            else VG_HIP(hipMemcpy(d.d_obs, d.h_obs.data(), sizeof(double) * nb * 2 * d.N, hipMemcpyHostToDevice));
            VG_HIP(hipMemcpy(d.d_seq, d.h_seq.data(), sizeof(int32_t) * nb, hipMemcpyHostToDevice));
        }
        // host copies are no longer needed; everything stays resident in HBM
        std::vector<double>().swap(d.h_obs);
    }
    // descriptors of the merged chain-prep launch
    std::vector<vg::PrepDataset> prep;
    int64_t first = 0;
    for (auto &d : p->dss) {
        if (!d.n_blocks) continue;
        vg::PrepDataset pd;
        pd.chain = d.chain;
        // image b uses element b of its sequence (the common case): no index array -> one dependent load less
        bool identity = true;
        for (size_t i = 0; i < d.h_seq.size() && identity; i++) identity = d.h_seq[i] == (int32_t)i;
        d.seq_identity = identity;
        pd.seq_index = identity ? nullptr : d.d_seq;
        pd.frames = d.d_frames;
        pd.first = first;
        pd.count = d.n_blocks;
        pd.frame_stride_d = d.frame_stride;
        prep.push_back(pd);
        first += d.n_blocks;
    }
    p->prep = prep;  // up to kPrepMax datasets travel by value in the arguments of vg_chain_prep_multi_kernel
    p->prep_blocks = first;
    if (prep.size() > (size_t)vg::kPrepMax) {  // more: one launch over a descriptor table in global memory
        VG_HIP(hipMalloc(&p->d_prep, sizeof(vg::PrepDataset) * prep.size()));
        VG_HIP(hipMemcpy(p->d_prep, prep.data(), sizeof(vg::PrepDataset) * prep.size(), hipMemcpyHostToDevice));
    }
    p->finalized = true;
    return VG_OK;
}

int64_t vg_problem_num_parameters(const vg_problem *p) { return p && p->finalized ? p->n_params : -1; }

int64_t vg_problem_camera_offset(const vg_problem *p, int camera_id)
{
    if (!p || !p->finalized || camera_id < 0 || camera_id >= (int)p->cams.size()) return -1;
    return p->cams[camera_id].offset;
}

int64_t vg_problem_transform_offset(const vg_problem *p, int transform_id, int64_t index)
{
    if (!p || !p->finalized || transform_id < 0 || transform_id >= (int)p->tfs.size()) return -1;
    const Transform &t = p->tfs[transform_id];
    if (index < 0 || index >= t.count) return -1;
    return t.offset + 6 * index;
}

int vg_problem_set_parameters(vg_problem *p, const double *host_params)
{
    if (!p || !host_params) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    VG_HIP(hipSetDevice(p->device));
    VG_HIP(hipMemcpyAsync(p->d_params, host_params, sizeof(double) * (size_t)p->n_params, hipMemcpyHostToDevice, p->stream));
    VG_HIP(hipStreamSynchronize(p->stream));
    p->frames_stale = true;
    return VG_OK;
}

int vg_problem_get_parameters(vg_problem *p, double *host_params)
{
    if (!p || !host_params) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    VG_HIP(hipSetDevice(p->device));
    VG_HIP(hipMemcpyAsync(host_params, p->d_params, sizeof(double) * (size_t)p->n_params, hipMemcpyDeviceToHost, p->stream));
    VG_HIP(hipStreamSynchronize(p->stream));
    return VG_OK;
}

double *vg_problem_parameters_device(vg_problem *p) { return p && p->finalized ? p->d_params : nullptr; }

int vg_problem_num_datasets(const vg_problem *p) { return p ? (int)p->dss.size() : -1; }
int64_t vg_dataset_num_blocks(const vg_problem *p, int d) { return valid_dataset(p, d) == VG_OK ? p->dss[d].n_blocks : -1; }
int vg_dataset_single_launch(const vg_problem *p, int d)
{
    return valid_dataset(p, d) == VG_OK && p->finalized ? (single_launch_dataset(p, p->dss[d]) ? 1 : 0) : -1;
}
int vg_dataset_num_points(const vg_problem *p, int d) { return valid_dataset(p, d) == VG_OK ? p->dss[d].N : -1; }
int vg_dataset_chain_len(const vg_problem *p, int d) { return valid_dataset(p, d) == VG_OK ? p->dss[d].L : -1; }
int vg_dataset_num_intrinsics(const vg_problem *p, int d)
{
    return valid_dataset(p, d) == VG_OK ? p->cams[p->dss[d].camera].K : -1;
}

}  // extern "C"

int vgi::ensure_frames(vg_problem *p)
{
    if (!p->frames_stale) return VG_OK;
    const int rc = vgi::prepare_at(p, p->d_params);
    if (rc == VG_OK) p->frames_stale = false;
    return rc;
}

int vgi::prepare_at(vg_problem *p, const double *d_params)
{
    // whatever point this is, the frames no longer belong to an earlier vg_problem_prepare
    p->frames_stale = d_params != p->d_params;
    if (!p->prep_blocks) return VG_OK;
    if (p->d_prep) {
        const unsigned int grid = (unsigned int)((p->prep_blocks + 63) / 64);
        hipLaunchKernelGGL(vg::vg_chain_prep_table_kernel, dim3(grid), dim3(64), 0, p->stream, d_params, (const vg::PrepDataset *)p->d_prep,
                           (int)p->prep.size(), (long long)p->prep_blocks);
        VG_HIP(hipGetLastError());
        return VG_OK;
    }
    for (size_t g0 = 0; g0 < p->prep.size(); g0 += vg::kPrepMax) {
        vg::PrepMultiArgs m;
        m.n = (int)(p->prep.size() - g0 < (size_t)vg::kPrepMax ? p->prep.size() - g0 : (size_t)vg::kPrepMax);
        unsigned int waves = 0;
        int widest = 0;
        for (int k = 0; k < m.n; k++) {
            m.ds[k] = p->prep[g0 + (size_t)k];
            m.first_wave[k] = waves;
            waves += (unsigned int)((m.ds[k].count + 63) / 64);
            widest = m.ds[k].frame_stride_d > widest ? m.ds[k].frame_stride_d : widest;
        }
        for (int k = m.n; k <= vg::kPrepMax; k++) m.first_wave[k] = waves;
        m.staged = waves >= vg::kPrepStagedMinWaves ? 1 : 0;
        hipLaunchKernelGGL(vg::vg_chain_prep_multi_kernel, dim3(waves), dim3(64), m.staged ? (size_t)64 * widest * sizeof(double) : (size_t)0, p->stream, d_params, m);
        VG_HIP(hipGetLastError());
    }
    return VG_OK;
}

extern "C" {

int vg_problem_force_prepared_frames(vg_problem *p, int on)
{
    if (!p) return fail(VG_ERR_INVALID_ARGUMENT, "problem is NULL");
    p->force_prepared_frames = on != 0;
    p->frames_stale = true;
    return VG_OK;
}

int vg_problem_prepare(vg_problem *p)
{
    if (!p) return fail(VG_ERR_INVALID_ARGUMENT, "problem is NULL");
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    VG_HIP(hipSetDevice(p->device));
    // Lazy: the frames are rebuilt by the first consumer that reads them from HBM (ensure_frames); an evaluation of
    // a single-member DIRECT chain derives them inside the emit kernel and never needs this launch.
    p->frames_stale = true;
    return VG_OK;
}

int vg_dataset_evaluate(vg_problem *p, int dataset_id, double *residuals, double *jac_intr, double *const *jac_member)
{
    int rc = valid_dataset(p, dataset_id);
    if (rc != VG_OK) return rc;
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    Dataset &d = p->dss[dataset_id];
    d.epoch = (d.epoch + 1) & 0xFFFFFFull;
    if (d.epoch == 0) d.epoch = 1;
    if (!d.n_blocks) return VG_OK;  // nothing to evaluate (outputs may be zero-sized / NULL)
    if (!residuals) return fail(VG_ERR_INVALID_ARGUMENT, "residuals is NULL");
    VG_HIP(hipSetDevice(p->device));
    const Camera &cam = p->cams[d.camera];
    bool want_jac = jac_intr != nullptr;
    for (int l = 0; l < d.L; l++)
        if (jac_member && jac_member[l]) want_jac = true;
    // one DIRECT member: the emit kernel walks the (trivial) chain itself -- one launch per evaluation (only while the
    // launch's output fits the Infinity Cache, see inline_chain_max_bytes).  The route is a function of the PROBLEM
    // alone (vg_dataset_single_launch), never of what ran before: the same parameters always give the same bits.
    const bool inline_chain = single_launch_dataset(p, d);
    if (!inline_chain && (rc = vgi::ensure_frames(p)) != VG_OK) return rc;

    // 32-bit observation indices inside a launch: chunk very large datasets by whole images
    int64_t max_obs = (int64_t)1 << 30;
    if (const long long h = vgi::debug_hook(vgi::kHookMaxObsPerLaunch)) max_obs = h > 0 ? h : max_obs;  // test hook for the chunked path
    const int64_t max_blocks_per_launch = max_obs / d.N > 0 ? max_obs / d.N : 1;
    for (int64_t b0 = 0; b0 < d.n_blocks; b0 += max_blocks_per_launch) {
        const int64_t nb = d.n_blocks - b0 < max_blocks_per_launch ? d.n_blocks - b0 : max_blocks_per_launch;
        vg::EmitArgs a;
        fill_emit_args(p, d, a, b0, nb, residuals, jac_intr, jac_member);
        switch (cam.model) {
        case VG_MODEL_EUCM: rc = launch_emit<vg::kEUCM>(p->stream, a, want_jac, inline_chain); break;
        case VG_MODEL_UCM: rc = launch_emit<vg::kUCM>(p->stream, a, want_jac, inline_chain); break;
        default: rc = launch_emit<vg::kMEI>(p->stream, a, want_jac, inline_chain); break;
        }
        if (rc != VG_OK) return rc;
    }
    return VG_OK;
}

}  // extern "C"

#include "vg_host_route.hpp"

extern "C" {

int vg_problem_evaluate(vg_problem *p, const vg_dataset_outputs *outs)
{
    if (!p || !outs) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    VG_HIP(hipSetDevice(p->device));
    const int n_ds = (int)p->dss.size();
    int rc;
    // which datasets can share a launch: every Jacobian-carrying evaluation whose frames fit the LDS and whose
    // observation count fits one launch; the rest (cost-only calls, huge sets) go through vg_dataset_evaluate
    std::vector<int> shared, alone;
    bool need_frames = false;
    for (int i = 0; i < n_ds; i++) {
        Dataset &d = p->dss[i];
        if (!d.n_blocks) {
            d.epoch = (d.epoch + 1) & 0xFFFFFFull;
            if (d.epoch == 0) d.epoch = 1;
            continue;
        }
        if (!outs[i].residuals) return fail(VG_ERR_INVALID_ARGUMENT, "residuals is NULL");
        bool want_jac = outs[i].jac_intr != nullptr;
        for (int l = 0; l < d.L; l++) want_jac = want_jac || outs[i].jac_member[l] != nullptr;
        if (want_jac && emit_frames_in_lds(d.N, d.frame_stride) && d.n_blocks * (int64_t)d.N < ((int64_t)1 << 30)) shared.push_back(i);
        else alone.push_back(i);
    }
    if (shared.size() < 2) {  // nothing to merge
        alone.insert(alone.end(), shared.begin(), shared.end());
        shared.clear();
    }
    if (vgi::debug_hook(vgi::kHookEmitEqualTiles) != 4)   // widest rows first: every die ends on its lightest tiles (hook 4: problem order)
        std::stable_sort(shared.begin(), shared.end(), [&](int a2, int b2) {
            const Dataset &da = p->dss[a2], &db = p->dss[b2];
            return p->cams[da.camera].K + 6 * da.L > p->cams[db.camera].K + 6 * db.L;
        });
    for (int i : shared) need_frames = need_frames || !single_launch_dataset(p, p->dss[i]);
    if (need_frames && (rc = vgi::ensure_frames(p)) != VG_OK) return rc;
    for (size_t g0 = 0; g0 < shared.size(); g0 += vg::kEmitMultiMax) {
        vg::EmitMultiArgs m;
        m.n = (int)(shared.size() - g0 < (size_t)vg::kEmitMultiMax ? shared.size() - g0 : (size_t)vg::kEmitMultiMax);
        unsigned int tiles = 0;
        size_t lds = 0;
        for (int k = 0; k < m.n; k++) {
            Dataset &d = p->dss[shared[g0 + k]];
            const vg_dataset_outputs &o = outs[shared[g0 + k]];
            d.epoch = (d.epoch + 1) & 0xFFFFFFull;
            if (d.epoch == 0) d.epoch = 1;
            fill_emit_args(p, d, m.ds[k], 0, d.n_blocks, o.residuals, o.jac_intr, o.jac_member);
            m.model[k] = p->cams[d.camera].model;
            m.inline_chain[k] = single_launch_dataset(p, d) ? 1 : 0;
            m.first_tile[k] = tiles;
            tiles += (m.ds[k].n_obs + vg::kEmitThreads - 1) / vg::kEmitThreads;
            const size_t need = emit_lds_bytes<vg::kEUCM>(true, true, d.N, d.frame_stride);  // same tile size for every model
            lds = need > lds ? need : lds;
        }
        for (int k = m.n; k <= vg::kEmitMultiMax; k++) m.first_tile[k] = tiles;
        {   // one store policy for the whole launch: its datasets share the Infinity Cache
            int64_t launch_bytes = 0;
            for (int k = 0; k < m.n; k++) launch_bytes += emit_output_bytes(m.ds[k], p->cams[p->dss[shared[g0 + k]].camera].K);
            for (int k = 0; k < m.n; k++) {
                m.ds[k].nt_stores = launch_bytes >= emit_nt_min_bytes() ? 1 : 0;
                m.ds[k].map_window = emit_map_window(launch_bytes);
            }
        }
        // pieces of equal bytes per XCD: a tile of dataset k weighs its bytes per observation -- once the pass is large
        // enough to be bound by the write stream (past the 256 MiB Infinity Cache); a small pass (a stereo pair: 104 MB in
        // 21 us) is bound by the latency of a tile, the same for every dataset, and keeps equal counts (measured: 20.9 us
        // against 22.5 us with byte weights)
        double w[vg::kEmitMultiMax], total = 0., bytes = 0.;
        for (int k = 0; k < m.n; k++) {
            const Dataset &d = p->dss[shared[g0 + k]];
            w[k] = 32. + 16. * (p->cams[d.camera].K + 6 * d.L);
            bytes += w[k] * m.ds[k].n_obs;
        }
        for (int k = 0; k < m.n; k++) {
            const long long hook = vgi::debug_hook(vgi::kHookEmitEqualTiles);
            if (hook == 1 || (hook != 2 && bytes < 256. * 1024 * 1024)) w[k] = 1.;
            total += w[k] * (m.first_tile[k + 1] - m.first_tile[k]);
        }
        unsigned int cut[9], longest = 0;
        cut[0] = 0;
        cut[8] = tiles;
        for (int x = 1; x < 8; x++) {
            const double target = total * x / 8.;
            double cum = 0.;
            unsigned int t = tiles;
            for (int k = 0; k < m.n; k++) {
                const unsigned int nk = m.first_tile[k + 1] - m.first_tile[k];
                if (cum + w[k] * nk >= target) {
                    t = m.first_tile[k] + (unsigned int)((target - cum) / w[k] + 0.5);
                    break;
                }
                cum += w[k] * nk;
            }
            cut[x] = t < cut[x - 1] ? cut[x - 1] : (t > tiles ? tiles : t);
        }
        for (int x = 0; x < 8; x++) {
            m.xcd_first[x] = cut[x];
            m.xcd_count[x] = cut[x + 1] - cut[x];
            longest = m.xcd_count[x] > longest ? m.xcd_count[x] : longest;
        }
        // default: XCD x takes the x-th eighth of EVERY dataset, dataset after dataset -- equal bytes and equal arithmetic per die
        // whatever the mix of models, and the datasets that need no prepared frames come first on every die (same box:
        // stereo 21.1 -> 19.6 us, rig 116 (equal tile counts) / 108.5 (equal bytes) / 109.4 us; hooks 1 / 2 keep the cut pieces)
        m.per_dataset = (vgi::debug_hook(vgi::kHookEmitEqualTiles) == 1 || vgi::debug_hook(vgi::kHookEmitEqualTiles) == 2) ? 0 : 1;
        if (m.per_dataset) {
            longest = 0;
            for (int k = 0; k < m.n; k++) longest += (m.first_tile[k + 1] - m.first_tile[k] + 7) / 8;
        }
        hipLaunchKernelGGL(vg::vg_emit_multi_kernel, dim3(8 * longest), dim3(vg::kEmitThreads), lds, p->stream, m);
        VG_HIP(hipGetLastError());
    }
    for (int i : alone)
        if ((rc = vg_dataset_evaluate(p, i, outs[i].residuals, outs[i].jac_intr, outs[i].jac_member)) != VG_OK) return rc;
    return VG_OK;
}

int vg_problem_synchronize(vg_problem *p)
{
    if (!p) return fail(VG_ERR_INVALID_ARGUMENT, "problem is NULL");
    VG_HIP(hipSetDevice(p->device));
    VG_HIP(hipStreamSynchronize(p->stream));
    return VG_OK;
}

int vg_dataset_failed_count(vg_problem *p, int dataset_id, int64_t *count)
{
    int rc = valid_dataset(p, dataset_id);
    if (rc != VG_OK) return rc;
    if (!count) return fail(VG_ERR_INVALID_ARGUMENT, "count is NULL");
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    VG_HIP(hipSetDevice(p->device));
    unsigned long long v = 0;
    VG_HIP(hipMemcpyAsync(&v, p->dss[dataset_id].d_failed, sizeof v, hipMemcpyDeviceToHost, p->stream));
    VG_HIP(hipStreamSynchronize(p->stream));
    const Dataset &d = p->dss[dataset_id];
    *count = (v >> 40) == d.epoch ? (int64_t)(v & ((1ull << 40) - 1)) : 0;
    return VG_OK;
}

/* ------------------------------------------------------------------------------------------ normal equations */

int vg_dataset_gram_width(const vg_problem *p, int d)
{
    if (valid_dataset(p, d) != VG_OK) return -1;
    return p->cams[p->dss[d].camera].K + 6 * p->dss[d].L + 1;
}


}  // extern "C"

extern "C" {


/* ------------------------------------------------------------------------------------------ per-block */

}  // extern "C"

#include "vg_block_group.hpp"

namespace {

int block_new(vg_block **out, int device, int model, int chain_len, const int *status, int n_points, const double *grid,
              const double *obs)
{
    if (!out) return fail(VG_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    const int K = vg::num_intrinsics(model);
    if (K < 0) return fail(VG_ERR_INVALID_ARGUMENT, "unknown camera model");
    if (chain_len < 0 || chain_len > VG_MAX_CHAIN) return fail(VG_ERR_INVALID_ARGUMENT, "chain length must be in [0, 5]");
    if (chain_len > 0 && !status) return fail(VG_ERR_INVALID_ARGUMENT, "status is NULL");
    if (n_points <= 0 || !grid || !obs) return fail(VG_ERR_INVALID_ARGUMENT, "empty block");
    for (int l = 0; l < chain_len; l++)
        if (status[l] != VG_TRANSFORM_DIRECT && status[l] != VG_TRANSFORM_INVERSE)
            return fail(VG_ERR_INVALID_ARGUMENT, "status must be DIRECT or INVERSE");
    int rc = check_device(device);
    if (rc != VG_OK) return rc;
    vg_block *b = new (std::nothrow) vg_block();
    if (!b) return fail(VG_ERR_ALLOC, "out of host memory");
    b->device = device;
    b->model = model;
    b->K = K;
    b->L = chain_len;
    b->N = n_points;
    for (int l = 0; l < chain_len; l++) b->status[l] = status[l];
    b->h_grid.assign(grid, grid + 3 * (size_t)n_points);
    b->h_obs.assign(obs, obs + 2 * (size_t)n_points);
    b->used.assign((size_t)K + 6 * (size_t)chain_len, 0.);
    *out = b;
    return VG_OK;
}

}  // namespace

// the block's own one-image problem and its staging buffers
int vgg::ensure_private(vg_block *b)
{
    if (b->p) return VG_OK;
    int rc = vg_problem_create(&b->p, b->device, nullptr);
    std::vector<double> zeros(VG_MAX_INTRINSICS, 0.);
    int cam = -1, ds = -1, tids[vg::kMaxChain] = {0};
    if (rc == VG_OK) rc = vg_problem_add_camera(b->p, b->model, zeros.data(), 0, &cam);
    for (int l = 0; l < b->L && rc == VG_OK; l++) rc = vg_problem_add_transform(b->p, 1, 0, 1, nullptr, &tids[l]);
    if (rc == VG_OK) rc = vg_problem_add_dataset(b->p, cam, b->L, tids, b->status, b->N, b->h_grid.data(), 1, nullptr, b->h_obs.data(), &ds);
    if (rc == VG_OK) rc = vg_problem_finalize(b->p);
    if (rc == VG_OK) {
        const size_t rows = 2 * (size_t)b->N;
        const size_t total = rows * (1 + (size_t)b->K + 6 * (size_t)b->L);
        hipError_t e = hipMalloc(&b->d_out, sizeof(double) * total);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&b->h_out), sizeof(double) * total, hipHostMallocDefault);
        if (e == hipSuccess)
            e = hipHostMalloc(reinterpret_cast<void **>(&b->h_params), sizeof(double) * ((size_t)b->K + 6 * (size_t)b->L),
                              hipHostMallocDefault);
        if (e != hipSuccess) rc = fail(VG_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
        else {
            b->d_res = b->d_out;
            b->d_jintr = b->d_out + rows;
            for (int l = 0; l < b->L; l++) b->d_jm[l] = b->d_out + rows * (1 + (size_t)b->K + 6 * (size_t)l);
        }
    }
    return rc;
}

namespace {

// one block on its own: H2D of its parameters, evaluation, D2H, synchronisation
int block_evaluate_alone(vg_block *b, double const *const *parameters, double *residuals, double **jacobians)
{
    int rc = vgg::ensure_private(b);
    if (rc != VG_OK) return rc;
    // parameter vector layout of the one-image problem: [intrinsics | member 0 | member 1 ...]
    std::memcpy(b->h_params, parameters[0], sizeof(double) * b->K);
    for (int l = 0; l < b->L; l++) std::memcpy(b->h_params + b->K + 6 * l, parameters[1 + l], sizeof(double) * 6);
    vg_problem *p = b->p;
    VG_HIP(hipSetDevice(p->device));
    hipStream_t s = p->stream;
    VG_HIP(hipMemcpyAsync(p->d_params, b->h_params, sizeof(double) * (size_t)p->n_params, hipMemcpyHostToDevice, s));
    p->frames_stale = true;  // new parameters: vg_dataset_evaluate rebuilds the frames (in-kernel for a single DIRECT member)
    double *jm[vg::kMaxChain] = {nullptr};
    double *ji = nullptr;
    size_t last = 2 * (size_t)b->N;  // doubles to bring back: up to the end of the last requested block
    const size_t rows = 2 * (size_t)b->N;
    if (jacobians) {
        if (jacobians[0]) {
            ji = b->d_jintr;
            last = rows * (1 + (size_t)b->K);
        }
        for (int l = 0; l < b->L; l++)
            if (jacobians[1 + l]) {
                jm[l] = b->d_jm[l];
                last = rows * (1 + (size_t)b->K + 6 * (size_t)(l + 1));
            }
    }
    rc = vg_dataset_evaluate(p, 0, b->d_res, ji, jm);
    if (rc != VG_OK) return rc;
    VG_HIP(hipMemcpyAsync(b->h_out, b->d_out, sizeof(double) * last, hipMemcpyDeviceToHost, s));
    VG_HIP(hipStreamSynchronize(s));
    std::memcpy(residuals, b->h_out, sizeof(double) * rows);
    if (ji) std::memcpy(jacobians[0], b->h_out + rows, sizeof(double) * rows * b->K);
    for (int l = 0; l < b->L; l++)
        if (jm[l]) std::memcpy(jacobians[1 + l], b->h_out + rows * (1 + (size_t)b->K + 6 * (size_t)l), sizeof(double) * rows * 6);
    return VG_OK;
}

void group_unseal(vg_block_group *g)
{
    if (g->p) vg_problem_destroy(g->p);
    g->p = nullptr;
    (void)hipSetDevice(g->device);
    if (g->d_out) (void)hipFree(g->d_out);
    if (g->h_mirror) (void)hipHostFree(g->h_mirror);
    if (g->h_params) (void)hipHostFree(g->h_params);
    g->d_out = g->h_mirror = g->h_params = nullptr;
    g->dss.clear();
    g->sealed = false;
    g->cooldown = 0;
    for (vg_block *b : g->blocks) b->used_valid = false;
}

}  // namespace

extern "C" {

int vg_block_create(vg_block **out, int device, int model, int chain_len, const int *status, int n_points,
                    const double *grid, const double *obs)
{
    int rc = block_new(out, device, model, chain_len, status, n_points, grid, obs);
    if (rc != VG_OK) return rc;
    if ((rc = vgg::ensure_private(*out)) != VG_OK) {
        vg_block_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

int vg_block_group_create(vg_block_group **out, int device, int mode)
{
    if (!out) return fail(VG_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (mode != VG_GROUP_IN_PLACE && mode != VG_GROUP_STATE_VECTOR) return fail(VG_ERR_INVALID_ARGUMENT, "unknown group mode");
    int rc = check_device(device);
    if (rc != VG_OK) return rc;
    vg_block_group *g = new (std::nothrow) vg_block_group();
    if (!g) return fail(VG_ERR_ALLOC, "out of host memory");
    g->device = device;
    g->mode = mode;
    *out = g;
    return VG_OK;
}

int vg_block_create_in_group(vg_block **out, vg_block_group *g, int model, int chain_len, const int *status, int n_points,
                             const double *grid, const double *obs)
{
    if (!g) return fail(VG_ERR_INVALID_ARGUMENT, "group is NULL");
    int rc = block_new(out, g->device, model, chain_len, status, n_points, grid, obs);
    if (rc != VG_OK) return rc;
    if (g->sealed) group_unseal(g);  // a new member: the resident problem is rebuilt once it has been seen
    (*out)->group = g;
    g->blocks.push_back(*out);
    return VG_OK;
}

int vg_block_group_stats(const vg_block_group *g, int64_t *n_blocks, int64_t *batched_evaluations, int64_t *served, int64_t *alone)
{
    if (!g) return fail(VG_ERR_INVALID_ARGUMENT, "group is NULL");
    if (n_blocks) *n_blocks = (int64_t)g->blocks.size();
    if (batched_evaluations) *batched_evaluations = (int64_t)g->n_batched;
    if (served) *served = (int64_t)g->n_served;
    if (alone) *alone = (int64_t)g->n_alone;
    return VG_OK;
}

int vg_block_group_invalidate(vg_block_group *g)
{
    if (!g) return fail(VG_ERR_INVALID_ARGUMENT, "group is NULL");
    g->seen.clear();
    g->n_known = 0;
    g->n_stale = 0;
    g->cooldown = 0;
    for (vg_block *b : g->blocks) {
        b->used_valid = false;
        b->calls = 0;
        for (int k = 0; k <= b->L; k++) b->moves[k] = false;
        if (b->is_bound) {
            b->stale = true;
            g->n_stale++;
        }
    }
    return VG_OK;
}

void vg_block_group_destroy(vg_block_group *g)
{
    if (!g) return;
    group_unseal(g);
    for (vg_block *b : g->blocks) b->group = nullptr;  // surviving blocks fall back to the per-block path
    delete g;
}

int vg_block_num_residuals(const vg_block *b) { return b ? 2 * b->N : -1; }
int vg_block_num_parameter_blocks(const vg_block *b) { return b ? 1 + b->L : -1; }
int vg_block_parameter_block_size(const vg_block *b, int idx)
{
    if (!b || idx < 0 || idx > b->L) return -1;
    return idx == 0 ? b->K : 6;
}

int vg_block_evaluate(vg_block *b, double const *const *parameters, double *residuals, double **jacobians)
{
    if (!b || !parameters || !residuals) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    for (int i = 0; i <= b->L; i++)
        if (!parameters[i]) return fail(VG_ERR_INVALID_ARGUMENT, "NULL parameter block");
    vg_block_group *g = b->group;
    if (!g) return block_evaluate_alone(b, parameters, residuals, jacobians);
    bool want_jac = false;
    if (jacobians)
        for (int i = 0; i <= b->L; i++) want_jac = want_jac || jacobians[i] != nullptr;
    int rc;
    vgg::observe(g, b, parameters);
    if (g->sealed && g->n_stale > 0) {
        // after vg_block_group_invalidate: what the group knew about where parameters live is void; every block
        // evaluates alone once and is bound again before the next pass
        if ((rc = block_evaluate_alone(b, parameters, residuals, jacobians)) != VG_OK) return rc;
        g->n_alone++;
        vgg::bind(g, b, parameters);
        if (b->stale) {
            b->stale = false;
            g->n_stale--;
        }
        return VG_OK;
    }
    if (!g->sealed) {
        // first pass: every block is seen once on its own and bound to the pointers it was called with
        if ((rc = block_evaluate_alone(b, parameters, residuals, jacobians)) != VG_OK) return rc;
        g->n_alone++;
        vgg::bind(g, b, parameters);
        if (g->n_bound == (int)g->blocks.size() && (rc = vgg::seal(g)) != VG_OK) {
            group_unseal(g);  // the group stays usable, block by block
            g->n_bound = -1;  // ... and does not try again
            return VG_OK;
        }
        return VG_OK;
    }
    if (!(vgg::params_match(b, parameters) && (!want_jac || g->point_has_jac)) && vgg::worth_a_pass(g, b, parameters)) {
        // a new evaluation point (or the Jacobians of a point that so far only had its cost evaluated): one pass over
        // ALL blocks of the group, at the parameter values the other blocks are expected to be called with
        if ((rc = vgg::evaluate_all(g, b, parameters, want_jac)) != VG_OK) return rc;
    }
    if (vgg::params_match(b, parameters) && (!want_jac || g->point_has_jac)) {
        vgg::serve(g, b, residuals, jacobians);
        g->n_served++;
        g->served_since_batch++;
    } else {
        // the prediction of this block's parameters was wrong (a host that does not keep the layout the mode assumes)
        if ((rc = block_evaluate_alone(b, parameters, residuals, jacobians)) != VG_OK) return rc;
        g->n_alone++;
    }
    vgg::bind(g, b, parameters);
    return VG_OK;
}

void vg_block_destroy(vg_block *b)
{
    if (!b) return;
    if (b->group) {
        vg_block_group *g = b->group;
        if (g->sealed) group_unseal(g);
        g->blocks.erase(std::remove(g->blocks.begin(), g->blocks.end(), b), g->blocks.end());
        if (b->is_bound && g->n_bound > 0) g->n_bound--;
        if (b->calls >= 2 && g->n_known > 0) g->n_known--;
        if (b->stale && g->n_stale > 0) g->n_stale--;
    }
    (void)hipSetDevice(b->device);
    if (b->d_out) (void)hipFree(b->d_out);
    if (b->h_out) (void)hipHostFree(b->h_out);
    if (b->h_params) (void)hipHostFree(b->h_params);
    vg_problem_destroy(b->p);
    delete b;
}

/* ------------------------------------------------------------------------------------------ measurement helpers */

int vg_calib_stream_write(void *hip_stream, double *dst, int64_t n_doubles, double value)
{
    if (!dst || n_doubles < 0 || (n_doubles & 1)) return fail(VG_ERR_INVALID_ARGUMENT, "need an even number of doubles");
    const long long n2 = n_doubles / 2;
    const unsigned int grid = (unsigned int)((n2 + 256 * vg::kStreamUnroll - 1) / (256 * vg::kStreamUnroll));
    if (!grid) return VG_OK;
    hipLaunchKernelGGL(vg::vg_stream_write_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream),
                       dst, n2, value);
    VG_HIP(hipGetLastError());
    return VG_OK;
}

int vg_calib_fp64_fma(void *hip_stream, double *scratch, int iters, int64_t *flops_out)
{
    if (!scratch || iters < 1 || !flops_out) return fail(VG_ERR_INVALID_ARGUMENT, "scratch / iters / flops_out");
    int dev = 0;
    hipDeviceProp_t prop;
    VG_HIP(hipGetDevice(&dev));
    VG_HIP(hipGetDeviceProperties(&prop, dev));
    const size_t lds = 72 * 1024;   // two workgroups of four waves per CU = two waves per SIMD: the fused Gram kernels' occupancy
    static bool raised = false;
    if (!raised) {
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vg::vg_fp64_fma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = true;
    }
    const unsigned int grid = 2u * (unsigned int)prop.multiProcessorCount;
    hipLaunchKernelGGL(vg::vg_fp64_fma_kernel, dim3(grid), dim3(256), lds, reinterpret_cast<hipStream_t>(hip_stream), scratch, iters, 1.0);
    VG_HIP(hipGetLastError());
    *flops_out = (int64_t)grid * 256 * (int64_t)iters * vg::kFmaChains * 2;
    return VG_OK;
}

int vg_calib_d2h_copies(int device, int64_t bytes, int reps, double *seconds_out)
{
    if (bytes <= 0 || reps <= 0 || !seconds_out) return fail(VG_ERR_INVALID_ARGUMENT, "bad arguments");
    VG_HIP(hipSetDevice(device));
    void *dev = nullptr, *host = nullptr;
    VG_HIP(hipMalloc(&dev, (size_t)bytes));
    if (hipHostMalloc(&host, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipFree(dev);
        return fail(VG_ERR_ALLOC, "hipHostMalloc failed");
    }
    int rc = VG_OK;
    if (hipMemset(dev, 1, (size_t)bytes) != hipSuccess) rc = fail(VG_ERR_HIP, "hipMemset failed");
    std::memset(host, 0, (size_t)bytes);
    if (rc == VG_OK && hipMemcpy(host, dev, (size_t)bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(VG_ERR_HIP, "hipMemcpy failed");  // warm
    for (int r = 0; r < reps && rc == VG_OK; r++) {
        const auto t0 = std::chrono::steady_clock::now();
        if (hipMemcpy(host, dev, (size_t)bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(VG_ERR_HIP, "hipMemcpy failed");
        seconds_out[r] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    (void)hipHostFree(host);
    (void)hipFree(dev);
    return rc;
}

int vg_calib_stream_copy(void *hip_stream, double *dst, const double *src, int64_t n_doubles)
{
    if (!dst || !src || n_doubles < 0 || (n_doubles & 1)) return fail(VG_ERR_INVALID_ARGUMENT, "need an even number of doubles");
    const long long n2 = n_doubles / 2;
    const unsigned int grid = (unsigned int)((n2 + 256 * vg::kStreamUnroll - 1) / (256 * vg::kStreamUnroll));
    if (!grid) return VG_OK;
    hipLaunchKernelGGL(vg::vg_stream_copy_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream),
                       dst, src, n2);
    VG_HIP(hipGetLastError());
    return VG_OK;
}

}  // extern "C"

// The other translation units of the library (each owns its kernels; they meet through include/visgeom_amd.h and the
// vgi:: declarations of vg_internal.hpp): vg_solver_tu.hip (vg_comm.hpp, vg_solver_impl.hpp), vg_refine_tu.hip
// (vg_pose_lm.hpp, vg_refine_impl.hpp), vg_frontend_tu.hip (vg_calibration.hpp), vg_local_tu.hip (vg_local_impl.hpp).
