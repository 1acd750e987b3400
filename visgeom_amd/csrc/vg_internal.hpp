// vg_internal.hpp -- host-side state shared by the translation units of libvisgeom_amd.so (not installed).
#pragma once

#include "../../include/visgeom_amd.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "vg_kernels.hpp"
#include "vg_odometry.hpp"

namespace vgi {

int fail(int code, const std::string &msg);

// Measurement / test hooks (A/B switches of the kernels, the chunked-launch test): a process-wide table set through
// vg_debug_set() -- the library reads NO environment variable for them.  Built without VG_DEBUG_HOOKS (a production build:
// VG_PRODUCTION=1 python -m visgeom_amd._build) the table does not exist, every hook is its default at compile time and
// vg_debug_set() fails.
enum DebugHook {
    kHookInlineChainMaxBytes = 0,  // largest evaluation (bytes) whose single-member chain is walked in the emit kernel
    kHookGramForceMfma,            // single-member chains on the matrix-core Gram kernel as well
    kHookGramCh1,                  // one corner per lane in the vector-pipe Gram kernel
    kHookGramNoMerge,              // 1: one Gram launch per dataset; 2: merged launch in dataset order instead of heaviest first
    kHookMaxObsPerLaunch,          // chunk size of the emit launches (the chunked path without a 240 GB problem)
    kHookSolverTiming,             // print where the solver's set-up time goes
    kHookSolverHostLoop,           // force the host-driven LM loop
    kHookSolverDeviceLoop,         // force the device-resident LM loop
    kHookSolverNoSpeculation,      // queue one LM iteration at a time
    kHookEmitEqualTiles,           // merged emit launch: contiguous XCD pieces of 1 = equal tile counts, 2 = equal bytes (default: an eighth of every dataset, widest rows first; 4: in problem order)
    kHookSchurPrivateGather,       // Schur rows kernel: every lane of a pose gathers V_i / g_i itself (the route before round 4), for A/B
    kHookSolverEventWait,          // device-resident loop: wait for an event behind every accept kernel instead of spinning on its sequence word (A/B)
    kHookSolverNoFoldFrames,       // LM loops: launch the chain prep in front of every candidate evaluation instead of building the candidate's frames in the back-substitution kernel (A/B, bit-equality test)
    kHookSolverOneWaveFold,        // vg_backsub_solve_kernel: the reduced system by the first wave alone, a row per lane (the route before the entry-parallel L D L^T; A/B)
    kHookSolverFoldMaxGroups,      // largest number of back-substitution workgroups whose launch also solves the reduced system (each workgroup redundantly); beyond: a one-workgroup solve launch in front (0 = the default, kFoldMaxGroups)
    kHookEmitNtMinBytes,           // smallest launch output (bytes) written with non-temporal stores (0 = the default; 1 = always; a huge value = never)
    kHookHostChunkBytes,           // chunk size of vg_dataset_evaluate_to_host in bytes (0 = the default, 32 MiB): tests force many small chunks
    kHookGramStamps,               // measurement build (-DVG_GRAM_STAMPS) only: device address of the per-wave clock stamps of the Gram kernel
    kHookGramPersistent,           // persistent form of the direct Gram kernel (vg_gram_valu_pers_kernel): 1 = never, 2 / 3 = its four- / eight-wave shape whenever it applies, 0 = by size
    kHookEmitMapWindow,            // tile map of the emit launches: W > 0 = windows of 8 W tiles, XCD x the x-th run of W tiles in each (1 = linear map); -1 = one contiguous eighth per XCD (the map before round 6); 0 = the default (kEmitMapWindow)
    kHookCount
};
#ifdef VG_DEBUG_HOOKS
long long debug_hook(DebugHook h);  // 0 = unset
#else
constexpr long long debug_hook(DebugHook) { return 0; }
#endif

#define VG_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            return vgi::fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? VG_ERR_NO_DEVICE \
                                                                                   : VG_ERR_HIP,  \
                             std::string(#expr) + ": " + hipGetErrorString(e_));                  \
        }                                                                                         \
    } while (0)

struct Camera {
    int model = 0, K = 0;
    bool constant = false;
    int64_t offset = -1;
    std::vector<double> init;
};

struct Transform {
    bool global = true, constant = false;
    int64_t count = 1;
    int64_t offset = -1;
    std::vector<double> init;
};

// a free-standing global parameter block (the wheel radii / track gauge of OdometryCost, unified_calibration.cpp:680-683)
struct ParamBlock {
    int size = 0;
    bool constant = false;
    int64_t offset = -1;
    std::vector<double> init;
};

// The corners of one dataset in HBM, uploaded ONCE and shared by everything that reads them: the per-image refinement of
// estimateInitialGrid, the initGlobalTransform sub-problem and the global problem (vg_calibration.hpp; reference flow
// unified_calibration.cpp:1137-1155 then :514-630).  Freed with the last owner.
struct CornerBlock {
    int device = 0;
    double *d_obs = nullptr;   // [n_images][N][2]
    int64_t n_images = 0;
    int N = 0;
    CornerBlock() = default;
    CornerBlock(const CornerBlock &) = delete;
    CornerBlock &operator=(const CornerBlock &) = delete;
    ~CornerBlock();
};

struct Dataset {
    int camera = -1, L = 0, N = 0;
    int tids[vg::kMaxChain] = {0};
    int status[vg::kMaxChain] = {0};
    int64_t n_blocks = 0;
    std::vector<double> h_board, h_obs;
    std::vector<int32_t> h_seq;
    bool seq_identity = true;  // image b uses element b of its sequence: no index array needed on the device
    bool zero_obs = false;     // added without corners: the observations are zeros (cleared on the device, nothing uploaded)
    std::shared_ptr<CornerBlock> resident;   // the observations live in a shared block (d_obs points into it, not owned)
    double *d_board = nullptr, *d_obs = nullptr, *d_frames = nullptr;
    int32_t *d_seq = nullptr;
    unsigned long long *d_failed = nullptr;
    // vg_dataset_evaluate_to_host (vg_host_route.hpp): the rows travel chunk by chunk -- device staging laid out chunk-major
    // [res | jac_intr | jac_member ...] per chunk, a copy stream of its own, one event pair per chunk, and (for destinations
    // that are not pinned) a pinned staging block of the same layout that the host's threads scatter into the caller's arrays
    double *d_host_stage = nullptr, *h_host_stage = nullptr;
    size_t d_host_stage_doubles = 0, h_host_stage_doubles = 0;
    hipStream_t host_copy_stream = nullptr;
    std::vector<hipEvent_t> host_chunk_ready, host_chunk_copied;
    double *d_partials = nullptr;  // [ceil(n_blocks / kSlab)][W*W] workspace of vg_dataset_gram_sum
    double *d_wg_partials = nullptr;  // [W(W+1)/2][n_workgroups] per-workgroup sums of the vector-pipe Gram kernel
    unsigned long long epoch = 0;  // evaluation counter, tags d_failed
    int frame_stride = 0;
    vg::ChainDesc chain;
};

// TransformationPrior block (include/calibration/calib_cost_functions.h:79-103, .cpp:214-228) on a global transform
struct Prior {
    int tf = -1;
    double A[36];  // row-major; diag(stiffness) with the rotation part multiplied by interOmegaRot(prior rot)
    double R[9];   // rotMat of the prior
    double xi[6];  // the prior value
};

}  // namespace vgi

struct vg_problem {
    int device = 0;
    hipStream_t stream = nullptr;
    bool finalized = false;
    std::vector<vgi::Camera> cams;
    std::vector<vgi::Transform> tfs;
    std::vector<vgi::Dataset> dss;
    std::vector<vgi::ParamBlock> pblocks;
    std::vector<vgi::Prior> priors;
    std::vector<vgodo::Block> odoms;                       // OdometryPrior blocks (consecutive elements of a sequence)
    std::vector<std::pair<int, int64_t>> const_poses;     // (sequence transform, index) held constant ("anchor")
    int64_t n_params = 0;
    double *d_params = nullptr;
    std::vector<vg::PrepDataset> prep;  // one descriptor per non-empty dataset (vg_chain_prep_multi_kernel takes them by value)
    vg::PrepDataset *d_prep = nullptr;  // the same as a table in global memory, only for problems of more than kPrepMax datasets
    int64_t prep_blocks = 0;
    // vg_problem_prepare marks the frames stale; they are rebuilt on demand (chain-prep kernel) by whoever reads
    // them from HBM -- or never, when every consumer derives them in-kernel (single-member DIRECT chains)
    bool frames_stale = true;
    // test / measurement hook (vg_problem_force_prepared_frames): every kernel reads the reference-order frames of the
    // chain-prep launch instead of walking single-member chains itself
    const int *gram_gate = nullptr;  // set by the solver around speculative launches (GramArgs::gate)
    int gram_gate_expect = 0;
    bool force_prepared_frames = false;
};

struct vg_block_group;

struct vg_block {
    vg_problem *p = nullptr;  // the block's own one-image problem (created on first use when the block is in a group)
    int device = 0;
    int model = 0, K = 0, L = 0, N = 0;
    int status[vg::kMaxChain] = {0};
    std::vector<double> h_grid, h_obs;  // kept for the private problem / the group's resident problem
    // one device allocation [res | jac_intr | jac_member 0 | ...] and one pinned host mirror of it: a call is one
    // H2D of the parameters, two launches, ONE D2H and one synchronisation
    double *d_out = nullptr, *h_out = nullptr;
    double *d_res = nullptr, *d_jintr = nullptr;
    double *d_jm[vg::kMaxChain] = {nullptr};
    double *h_params = nullptr;  // pinned, K + 6L doubles
    // ---- membership in a vg_block_group (vg_block_group.hpp)
    vg_block_group *group = nullptr;
    int g_ds = -1, g_idx = -1;
    const double *bound[1 + vg::kMaxChain] = {nullptr};  // parameter pointers of the last call
    bool moves[1 + vg::kMaxChain] = {false};             // ... that pointer has differed between two calls
    bool is_bound = false, used_valid = false;
    bool stale = false;                                  // bound before the last vg_block_group_invalidate, not called since
    int calls = 0;                                       // evaluations seen (saturates at 2: from then on `moves` is known)
    std::vector<double> used;                            // parameter values the group's last pass used for this block
};

namespace vgi {
inline int valid_dataset(const vg_problem *p, int d)
{
    if (!p) return fail(VG_ERR_INVALID_ARGUMENT, "problem is NULL");
    if (d < 0 || d >= (int)p->dss.size()) return fail(VG_ERR_INVALID_ARGUMENT, "dataset id out of range");
    return VG_OK;
}

// kernel launches on an explicit parameter buffer (the solver evaluates candidate points without
// touching the problem's own parameter vector); implemented in vg_capi.hip
int prepare_at(vg_problem *p, const double *d_params);
int ensure_frames(vg_problem *p);  // chain prep at the problem's own parameters if the frames are stale
// sum != NULL: also the fixed-order sum over the dataset's blocks, [W*W] (one extra launch on the vector-pipe route)
int gram_fused_at(vg_problem *p, int dataset_id, const double *d_params, double *gram, double *sum);
// the fused Gram of every dataset the merged vector-pipe launch can take (taken[d] = 1), at a device parameter buffer;
// grams[d] = that dataset's [n_blocks][W*W] output.  The others are the caller's (gram_fused_at).
int gram_fused_merged_at(vg_problem *p, const double *d_params, double *const *grams, std::vector<char> &taken,
                         double *const *partials /* per dataset [E][ceil(n_blocks / 8)] or NULL */);
bool gram_merge_covers_all(const vg_problem *p);  // every non-empty dataset goes through the merged launch
bool gram_needs_frames(const vg_problem *p);  // false when every dataset's Gram kernel walks its chain itself
bool gram_dataset_needs_frames(const vg_problem *p, int dataset_id);  // the same question for one dataset
int gram_sum_into(vg_problem *p, int dataset_id, const double *gram, double *sum);
// vg_refine_poses with a clock (vg_refine_impl.hpp): kernel_seconds (may be NULL) receives the duration of the
// vg_pose_lm_kernel launch alone, measured with HIP events on the launch stream
int refine_poses(int device, void *hip_stream, int model, const double *intrinsics, int n_points, const double *board, int64_t n_images,
                 const double *corners, double *poses, const vg_solve_options *options, int32_t *iterations, double *final_cost,
                 int32_t *termination, double *kernel_seconds);
// the same on observations that already live in HBM (d_obs [n_images][N][2]); intrinsics / board from device pointers or, when
// those are NULL, from the host arrays (uploaded with the poses in one copy).  locked: the caller holds the scratch mutex.
int refine_poses_resident(int device, void *hip_stream, int model, const double *d_intr, const double *h_intr, int n_points, const double *d_board,
                          const double *h_board, int64_t n_images, const double *d_obs, double *poses, const vg_solve_options *options,
                          int32_t *iterations, double *final_cost, int32_t *termination, double *kernel_seconds, bool locked = false);
void refine_release_cached();   // the refinement's cached device / pinned blocks (vg_release_cached_memory)
using GatherFn = std::function<void(int64_t first, int64_t count, double *dst)>;
int upload_corners(int device, void *hip_stream, int64_t n_images, int n_points, const GatherFn &gather, std::shared_ptr<CornerBlock> *out);
int problem_add_projection_dataset(vg_problem *p, int camera_id, int chain_len, const int *transform_ids, const int *status, int n_points,
                                   const double *board, int64_t n_images, const int32_t *image_index, int *dataset_id);
// vg_problem_add_dataset with the observations taken from a resident block (rows [0, n_images) of it) instead of a host array
int problem_add_dataset_resident(vg_problem *p, int camera_id, int chain_len, const int *transform_ids, const int *status, int n_points,
                                 const double *board, int64_t n_images, const int32_t *image_index, const std::shared_ptr<CornerBlock> &corners,
                                 int *dataset_id);
}  // namespace vgi
