/*
 * visgeom_amd.h -- C ABI of the MI355X-native reprojection residual / Jacobian engine.
 *
 * Drop-in boundary for ONE hot path of BKhomutenko/visgeom: the calibration cost function
 *     GenericProjectionJac::Evaluate        src/calibration/calib_cost_functions.cpp:28-117
 * and the normal-equation build (J^T J, J^T r) that Ceres performs on its output
 * (not in the reference tree; call sites src/calibration/unified_calibration.cpp:53,426,1152).
 *
 * Conventions
 *   - plain C, no torch / Eigen / Ceres types; every function returns a vg_status
 *     (VG_OK == 0) unless documented otherwise; vg_last_error() gives the text.
 *   - all arithmetic is IEEE double (the reference computes in double throughout).
 *   - "device pointer" = memory of the problem's HIP device (hipMalloc'ed or a torch tensor's
 *     data_ptr()); "host pointer" = ordinary memory.  Output buffers are caller-owned.
 *   - a 6-vector transform is [tx,ty,tz, rx,ry,rz] (translation, rotation vector),
 *     include/geometry/transformation.h:46.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry fails with
 *     VG_ERR_NO_DEVICE / VG_ERR_HIP.
 *
 * All file:line citations are relative to the reference tree (/root/reference).
 */
#ifndef VISGEOM_AMD_H
#define VISGEOM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VG_ABI_VERSION 1

/* camera models: parseCameras, src/calibration/unified_calibration.cpp:134-180 */
enum vg_model {
    VG_MODEL_EUCM = 0, /* [alpha,beta,fu,fv,u0,v0]          include/projection/eucm.h:29-226 */
    VG_MODEL_UCM = 1,  /* [xi,fu,fv,u0,v0]                  include/projection/ucm.h:32-197  */
    VG_MODEL_MEI = 2   /* [xi,k1,k2,k3,k4,k5,fu,fv,u0,v0]   include/projection/mei.h:29-285  */
};

/* enum TransformationStatus, include/calibration/calib_cost_functions.h:25 */
enum vg_transform_status { VG_TRANSFORM_DIRECT = 0, VG_TRANSFORM_INVERSE = 1 };

enum vg_status {
    VG_OK = 0,
    VG_ERR_INVALID_ARGUMENT = 1,
    VG_ERR_HIP = 2,       /* a HIP runtime call failed (text in vg_last_error) */
    VG_ERR_NO_DEVICE = 3, /* no HIP device: the product path has no CPU fallback */
    VG_ERR_STATE = 4,     /* call order violated (e.g. evaluate before finalize) */
    VG_ERR_ALLOC = 5,
    VG_ERR_NUMERIC = 6    /* solver: a factorisation failed */
};

#define VG_MAX_CHAIN 5       /* src/calibration/unified_calibration.cpp:566-567 (> 5 throws) */
#define VG_MAX_INTRINSICS 10 /* Mei */
#define VG_DOUBLE_BIG 1e15   /* include/std.h:71, in-band failed-projection residual */

int vg_abi_version(void);
const char *vg_last_error(void);       /* thread-local, never NULL */
int vg_device_count(void);             /* number of HIP devices, 0 when none / no driver */
int vg_num_intrinsics(int model);      /* 6 / 5 / 10, -1 for an unknown model */
/* box bounds of intrinsic idx: eucm.h:228-246, ucm.h:199-215, mei.h:287-313 */
int vg_intrinsic_bounds(int model, int idx, double *lower, double *upper);

/* =====================================================================================
 * 1. Per-block entry -- 1:1 with struct GenericProjectionJac
 *    (include/calibration/calib_cost_functions.h:27-62).  A 30-line adapter deriving from
 *    ceres::CostFunction forwards ctor / Evaluate / dtor to these three calls
 *    (INTEGRATION.md section 1).  One block = one image of one camera.
 * ===================================================================================== */
typedef struct vg_block vg_block;

/* ctor (calib_cost_functions.h:29-46): copies obs ("proj", 2N) and grid (3N), records the chain
 * statuses; parameter blocks are [K, 6 x chain_len], residual count 2N.  chain_len in [0,5]. */
int vg_block_create(vg_block **out, int device, int model, int chain_len, const int *status,
                    int n_points, const double *grid /*3N host*/, const double *obs /*2N host*/);
int vg_block_num_residuals(const vg_block *b);                  /* 2N        (:45) */
int vg_block_num_parameter_blocks(const vg_block *b);           /* 1 + L           */
int vg_block_parameter_block_size(const vg_block *b, int idx);  /* K or 6    (:38-42) */

/* Evaluate (calib_cost_functions.cpp:28-117).  Host pointers, exactly Ceres' contract:
 *   parameters[0] -> K intrinsics, parameters[1+l] -> 6-vector of chain member l;
 *   residuals[2N] = [u0,v0,u1,v1,...]; failed projection -> (1e15,1e15) and zero Jacobian rows;
 *   jacobians NULL, or 1+L pointers each NULL or row-major [2N x blocksize].
 * Returns VG_OK where the reference returns true (it always does, :116). */
int vg_block_evaluate(vg_block *b, double const *const *parameters, double *residuals,
                      double **jacobians);
void vg_block_destroy(vg_block *b);

/* ---- block groups: the same per-block entry, amortised.  Blocks created against a group share one resident problem:
 * the FIRST vg_block_evaluate at a new parameter point evaluates every block of the group in one pass (merged
 * launches, one D2H into a pinned mirror), every later call at that point compares its parameter values with the ones
 * the pass used (bitwise) and copies its rows out -- about 2 us per call at 10 000 blocks against ~45 us for a block
 * on its own and ~9 us for the reference's CPU Evaluate.  A block is never served rows computed from other parameter
 * values than the ones it passes: on a mismatch it evaluates alone.
 * The per-block interface shows a block only its own parameter pointers; where the OTHER blocks' parameters are when
 * a pass opens is learned from the first pass (every block evaluates alone once and is bound to its pointers; pointer
 * identity tells which blocks share intrinsics or a global transform) and then follows the mode:
 *   VG_GROUP_IN_PLACE      parameters are read where they were last seen (ceres::Problem::Evaluate, gradient
 *                          checkers, solvers that evaluate in place);
 *   VG_GROUP_STATE_VECTOR  the host evaluates candidate points held in state arrays of a fixed layout (ceres::Solve:
 *                          every variable parameter block of a pass is state + offset): a new pass is the previous
 *                          one displaced by the displacement of the calling block's pointers; blocks whose pointers
 *                          never move (constant parameter blocks) are read in place.  The host guarantees that the
 *                          A displaced address is only read when it lies inside memory the host has PASSED to some
 *                          block of the group before (the group keeps the address ranges it has been shown; the blocks of
 *                          one state array coalesce into one range), so a new state array costs one block-by-block pass
 *                          before it is batched, and pointer arithmetic never leaves memory the host has shown.
 * Pointers passed to vg_block_evaluate must stay readable until the block is called again OR the group is invalidated:
 * call vg_block_group_invalidate when that memory goes away -- in the drop-in, right after ceres::Solve returns (its state
 * arrays are freed) and before the report's Problem::Evaluate / Evaluate calls on user memory.  Not thread safe. */
typedef struct vg_block_group vg_block_group;
enum vg_group_mode { VG_GROUP_IN_PLACE = 0, VG_GROUP_STATE_VECTOR = 1 };
int vg_block_group_create(vg_block_group **out, int device, int mode);
/* vg_block_create with membership; blocks may be added at any time (the group re-learns its layout) */
int vg_block_create_in_group(vg_block **out, vg_block_group *g, int model, int chain_len, const int *status, int n_points,
                             const double *grid /*3N host*/, const double *obs /*2N host*/);
/* Forget where parameters live (bound pointers, shown address ranges, which pointers move): every block evaluates alone
 * once more and is bound again before the next pass.  The resident problem (which blocks form a dataset) is kept. */
int vg_block_group_invalidate(vg_block_group *g);
/* counters since creation: passes over the whole group / calls answered from a pass / calls evaluated alone */
int vg_block_group_stats(const vg_block_group *g, int64_t *n_blocks, int64_t *batched_evaluations, int64_t *served, int64_t *alone);
/* destroy the group after (or before) its blocks; surviving blocks continue on the per-block path */
void vg_block_group_destroy(vg_block_group *g);

/* =====================================================================================
 * 2. Batched problem -- what GenericCameraCalibration assembles
 *    (src/calibration/unified_calibration.cpp:91-180, 514-630), evaluated in ONE pass over
 *    all (image x corner) observations per dataset instead of one Evaluate per image.
 * ===================================================================================== */
typedef struct vg_problem vg_problem;

/* hip_stream: a hipStream_t to launch on (e.g. torch.cuda.current_stream().cuda_stream), or NULL
 * for the device's default stream.  All launches of this problem go to that stream, in order. */
int vg_problem_create(vg_problem **out, int device, void *hip_stream);
void vg_problem_destroy(vg_problem *p);

/* parseCameras (:134-180).  intrinsics: K host doubles.  constant: SetParameterBlockConstant (:614-617). */
int vg_problem_add_camera(vg_problem *p, int model, const double *intrinsics, int constant,
                          int *camera_id);
/* parseTransforms (:91-132).  is_global: one shared 6-vector; otherwise a sequence of `count`
 * 6-vectors indexed by image index (include/calibration/unified_calibration.h:161-165).
 * values: count*6 host doubles or NULL (zeros). */
int vg_problem_add_transform(vg_problem *p, int is_global, int constant, int count,
                             const double *values, int *transform_id);
/* addGridResidualBlocks (:514-630): one residual block per listed image.
 *   transform_ids/status: the chain, camera-side first (initTransformChainInfo :182-231);
 *     at most one member may be a sequence transform... the reference requires exactly one (:223-228),
 *     this ABI also accepts none (all-global chains) and chain_len 0.
 *   board: 3*n_points host doubles (initGrid :279-309 / initGridIR :234-250);
 *   image_index[n_images]: index into the sequence transform(s) for each block; images whose
 *     corner list was empty are simply not listed (:520);
 *   corners: [n_images][2*n_points] host doubles, [u0,v0,u1,v1,...] per image (NULL with n_images > 0 is an error). */
int vg_problem_add_dataset(vg_problem *p, int camera_id, int chain_len, const int *transform_ids,
                           const int *status, int n_points, const double *board, int64_t n_images,
                           const int32_t *image_index, const double *corners, int *dataset_id);
/* TransformationPrior (include/calibration/calib_cost_functions.h:79-103, .cpp:214-228; parseData :808-829): six
 * residuals A * [R e_t; R e_r], e = prior^-1 o xi, pulling a transform towards the value it has NOW (the reference
 * requires the transform to have a prior value and uses it).  stiffness: the 6 diagonal weights.  On a sequence
 * transform the block acts on element 0, as in the reference (getTransformData(name), :826).  Seen by
 * vg_problem_solve, not by the per-dataset evaluation entries. */
int vg_problem_add_transformation_prior(vg_problem *p, int transform_id, const double *stiffness);
/* OdometryPrior (include/calibration/calib_cost_functions.h:64-77, .cpp:119-212; parseData :743-807): six residuals
 * between elements `index` and `index + 1` of a SEQUENCE transform, built from the two odometry poses xi1, xi2 and
 * the relative error model (err_v, err_w, lambda).  These blocks couple consecutive poses: the solver eliminates such
 * a sequence as a block-tridiagonal system on the host (fine for the few-hundred-pose odometry sets the reference
 * targets; not available together with a multi-rank all-reduce). */
int vg_problem_add_odometry_prior(vg_problem *p, int transform_id, int64_t index, double err_v, double err_w,
                                  double lambda, const double *xi1, const double *xi2);
/* Host-only evaluation of one OdometryPrior block, exported so the block can be checked without a solve:
 * constructor arguments (err_v, err_w, lambda, odometry poses xi1_odom / xi2_odom) + the two current poses ->
 * residual[6], J1[36], J2[36] (row-major; either Jacobian may be NULL). */
int vg_odometry_prior_evaluate(double err_v, double err_w, double lambda, const double *xi1_odom, const double *xi2_odom,
                               const double *xi1, const double *xi2, double *residual, double *J1, double *J2);
/* A free-standing global parameter block (1..16 doubles) at the end of the parameter vector -- the
 * [radius_left, radius_right, track_gauge] of data type "odometry_intrinsic" (unified_calibration.cpp:680-683). */
int vg_problem_add_parameter_block(vg_problem *p, int size, const double *values, int constant, int *block_id);
int64_t vg_problem_parameter_block_offset(const vg_problem *p, int block_id);
/* OdometryCost (include/calibration/odometry_cost_function.h:33-57, src/calibration/odometry_cost_function.cpp:147-266;
 * parseData :661-742): six residuals between elements `index` and `index + 1` of a SEQUENCE transform and the three
 * odometry intrinsics of a differential drive, r = A (zeta_odo(intrinsics)^-1 o (xi1^-1 o xi2)) with zeta_odo the
 * chain of the interval's n_steps wheel increments delta_q [n_steps][2] = (left, right).  Parameter blocks (6, 6, 3),
 * as the reference ADDS the block to its problem (:732-735; the class itself declares one block).  The
 * weighting A is fixed from the block's values at the time of the call.  Like OdometryPrior these blocks are
 * evaluated on the host and make the pose system of the sequence block tridiagonal (single rank). */
int vg_problem_add_odometry_cost(vg_problem *p, int transform_id, int64_t index, double err_v, double err_w, double lambda,
                                 int n_steps, const double *delta_q, int param_block_id);
/* Host-only evaluation of one OdometryCost block: constructor arguments + the three parameter blocks ->
 * zeta_prior[6] (NULL ok), residual[6], J1[36], J2[36], J3[18] (6 x 3), row-major, any Jacobian may be NULL. */
int vg_odometry_cost_evaluate(double err_v, double err_w, double lambda, int n_steps, const double *delta_q, const double *intr_prior,
                              const double *xi1, const double *xi2, const double *intr, double *zeta_prior, double *residual,
                              double *J1, double *J2, double *J3);
/* SetParameterBlockConstant on ONE element of a sequence ("anchor": true, :803-806). */
int vg_problem_set_pose_constant(vg_problem *p, int transform_id, int64_t index);
/* freezes the layout, uploads everything, allocates per-block frames. */
int vg_problem_finalize(vg_problem *p);

/* ---- parameter vector: [camera 0 | camera 1 | ... | transform 0 (count x 6) | transform 1 ... | parameter blocks] ---- */
int64_t vg_problem_num_parameters(const vg_problem *p);
int64_t vg_problem_camera_offset(const vg_problem *p, int camera_id);
int64_t vg_problem_transform_offset(const vg_problem *p, int transform_id, int64_t index);
int vg_problem_set_parameters(vg_problem *p, const double *host_params);  /* H2D, stream-ordered + sync */
int vg_problem_get_parameters(vg_problem *p, double *host_params);        /* D2H + sync */
double *vg_problem_parameters_device(vg_problem *p); /* device pointer, valid until destroy */

/* ---- shape queries ---- */
int vg_problem_num_datasets(const vg_problem *p);
int64_t vg_dataset_num_blocks(const vg_problem *p, int dataset_id);
int vg_dataset_num_points(const vg_problem *p, int dataset_id);
int vg_dataset_chain_len(const vg_problem *p, int dataset_id);
/* 1 when an evaluation of this dataset after a parameter change is a single launch (one DIRECT chain member, output
 * of the launch within reach of the Infinity Cache), 0 when it is chain prep + emit, -1 on a bad id */
int vg_dataset_single_launch(const vg_problem *p, int dataset_id);
int vg_dataset_num_intrinsics(const vg_problem *p, int dataset_id);

/* ---- evaluation (asynchronous on the problem's stream; sync with vg_problem_synchronize) ----
 * vg_problem_prepare: declares the device parameters changed.  Every block's transform chain (the two chain walks
 *   of calib_cost_functions.cpp:32-46 and :76-92, the InterJacobian ctor jacobian.h:139-152) is composed into a
 *   per-block frame by kernel 1 the next time a kernel reads frames from memory (Gram kernels, multi-member chains);
 *   for a chain of ONE member used DIRECT (mono calibration) kernel 2 derives the frame itself and an evaluation is a
 *   single launch.  Call after every parameter change (vg_problem_set_parameters implies it).
 * vg_dataset_evaluate: kernel 2 -- one thread per (image, corner): residuals + Jacobian rows in the
 *   Ceres block layout, block after block:
 *     residuals  [n_blocks][2N]            jac_intr [n_blocks][2N][K]   (row-major)
 *     jac_member[l] [n_blocks][2N][6]      any of jac_intr / jac_member / jac_member[l] may be NULL
 *   (device pointers; jac_member itself is a HOST array of chain_len device pointers). */
int vg_problem_prepare(vg_problem *p);
/* Route selection is a function of the problem alone (see vg_dataset_single_launch): a chain of one DIRECT member is
 * walked inside the consuming kernel, from xi itself (the reference's rotvec -> quaternion -> rotvec round trip of
 * compose() is skipped: |dR| < 1e-15); everything else reads the reference-order frames of the chain-prep launch.  Results
 * are bitwise reproducible per route.  on != 0 forces the chain-prep route for every dataset (tests, A/B measurements). */
int vg_problem_force_prepared_frames(vg_problem *p, int on);
int vg_dataset_evaluate(vg_problem *p, int dataset_id, double *residuals, double *jac_intr,
                        double *const *jac_member);
/* Every dataset of the problem in ONE pass: what Ceres' evaluator does when it walks all residual blocks at a new
 * point (src/calibration/unified_calibration.cpp:53 -> calib_cost_functions.cpp:28 per block).  outs[d] holds the
 * device pointers vg_dataset_evaluate would get for dataset d (unused chain slots NULL).  Datasets are merged into
 * shared launches (up to 8 per launch, any mix of camera models and chains): a stereo pair or a rig is one emit launch
 * instead of one per camera.  Same results, bit for bit, as the per-dataset entry. */
typedef struct vg_dataset_outputs {
    double *residuals;
    double *jac_intr;
    double *jac_member[VG_MAX_CHAIN];
} vg_dataset_outputs;
int vg_problem_evaluate(vg_problem *p, const vg_dataset_outputs *outs /* [vg_problem_num_datasets] */);
int vg_problem_synchronize(vg_problem *p);
/* The same evaluation delivered to HOST memory (what a Ceres EvaluationCallback needs):
 * runs kernel 2 into library-owned device buffers, copies the Ceres-layout arrays to the given host pointers
 * (pinned memory recommended; any of the Jacobian pointers may be NULL) and synchronises.  Block b of the dataset
 * is then at  residuals + b*2N,  jac_intr + b*2N*K,  jac_member[l] + b*2N*6. */
int vg_dataset_evaluate_to_host(vg_problem *p, int dataset_id, double *residuals, double *jac_intr,
                                double *const *jac_member);

/* number of failed projections (1e15 residual pairs) seen by the last vg_dataset_evaluate of that
 * dataset; synchronises the stream.  Not in the reference. */
int vg_dataset_failed_count(vg_problem *p, int dataset_id, int64_t *count);

/* =====================================================================================
 * 3. Normal-equation build (what Ceres does with Evaluate's output; not in the reference tree:
 *    SURVEY section 0 fact 2, call sites src/calibration/unified_calibration.cpp:53,426,1152).
 *    Per residual block b:  G_b = S_b^T S_b  with  S_b = [J_0 | J_1 ... J_L | r]  (2N x W, W = K + 6L + 1).
 *    G_b holds J^T J (leading (W-1)x(W-1)), J^T r (last column) and r^T r (last entry) of the block, in
 *    the block's own column order [intrinsics, chain member 0, ..., chain member L-1, residual].
 * ===================================================================================== */
int vg_dataset_gram_width(const vg_problem *p, int dataset_id); /* W */
/* fused: evaluates residuals and Jacobian rows in registers and contracts them there, on the FP64 vector pipe (products per
 * lane, recursive-halving sum over the 32 lanes of an image; chains of two or more members through the factored form
 * J_l = [p | p hat(X)] F_l: the per-corner rows stay K + 7 wide, the W x W block is a per-image congruence); J is never
 * written to HBM.  gram: device [n_blocks][W*W], row-major, full symmetric.  Needs
 * vg_problem_prepare at the current parameters, like vg_dataset_evaluate. */
int vg_dataset_gram_fused(vg_problem *p, int dataset_id, double *gram);
/* the same per-block matrices AND their fixed-order sum over the dataset's blocks (sum[W*W] device, full symmetric)
 * in two launches: narrow row blocks (W <= 13: EUCM / UCM mono) are contracted on the FP64 vector pipe with the
 * chain walked in-kernel, each workgroup leaves the sum of its images, one final launch adds those.  Wider blocks:
 * vg_dataset_gram_fused + vg_dataset_gram_sum.  The sum equals vg_dataset_gram_sum's to rounding (different fixed
 * order) and is run-to-run reproducible. */
int vg_dataset_gram_fused_sum(vg_problem *p, int dataset_id, double *gram, double *sum);
/* vg_dataset_gram_fused for EVERY dataset of the problem (grams[d]: device [n_blocks][W*W], may be NULL for an empty
 * dataset): the datasets (chains of one to five members, boards of more than 32 points) share ONE launch -- a stereo pair
 * or a rig is several launches of a few hundred workgroups otherwise, each ending in a nearly empty round. */
int vg_problem_gram_fused(vg_problem *p, double *const *grams);
/* vg_problem_gram_fused AND the fixed-order sum of every dataset's blocks (sums[d]: device [W*W], required for every
 * dataset; an empty dataset's sum is zero): the merged launch leaves per-workgroup partial sums and ONE more launch adds
 * them for all datasets -- a normal-equation build of a stereo pair or a rig is chain prep + two launches.  Each sum is
 * bit-identical to vg_dataset_gram_fused_sum's. */
int vg_problem_gram_fused_sum(vg_problem *p, double *const *grams, double *const *sums);
/* two-pass: the same Gram matrices from rows already materialised by vg_dataset_evaluate
 * (all of residuals, jac_intr and every jac_member[l] are required). */
int vg_dataset_gram_from_rows(vg_problem *p, int dataset_id, const double *residuals, const double *jac_intr,
                              const double *const *jac_member, double *gram);
/* sum over the dataset's blocks in a fixed order (two-stage tree, no atomics): sum[W*W] device. */
int vg_dataset_gram_sum(vg_problem *p, int dataset_id, const double *gram, double *sum);

/* =====================================================================================
 * 4. Solve -- replaces ceres::Solve(options, &globalProblem, &summary) at
 *    src/calibration/unified_calibration.cpp:53 for problems made of GenericProjectionJac blocks.
 *    Levenberg-Marquardt (Ceres' trust-region policy), per-pose 6x6 blocks eliminated on the GPU
 *    (Schur complement), the small global system factored on the host.  Constant blocks stay fixed
 *    (:604-617); non-constant intrinsics are kept inside the camera's box bounds (:621-627).
 *    The result is written to the problem's parameter vector.
 * ===================================================================================== */
/* ---- multi-GPU: one process per GPU, images sharded over ranks (contiguous image ranges, both cameras of a frame on
 * the same rank), global parameters replicated.  The path has ONE exchange step -- the sum of the small normal-equation
 * blocks -- done by RCCL on DEVICE buffers, in place, on the problem's stream.  Not in the reference (single process,
 * SURVEY section 5); it serves the replacement of ceres::Solve at src/calibration/unified_calibration.cpp:53.
 * RCCL is bound at run time: nothing here is needed, or loaded, on one GPU. */
typedef struct vg_comm vg_comm;
#define VG_COMM_ID_BYTES 128
/* ncclGetUniqueId: rank 0 calls it, the host hands the bytes to every rank (MPI, torch.distributed, a file ...) */
int vg_comm_unique_id(char *id /* VG_COMM_ID_BYTES */);
/* ncclCommInitRank on `device`; collective over all n_ranks ranks */
int vg_comm_create(vg_comm **out, const char *id, int n_ranks, int rank, int device);
/* wrap an ncclComm_t the host application already owns (not destroyed by vg_comm_destroy) */
int vg_comm_adopt(vg_comm **out, void *nccl_comm, int device);
/* A communicator without RCCL behind it: this process stands for `replicas` ranks that all hold the SAME shard, so every
 * sum over ranks is `replicas` times the local value.  It lets a one-GPU box run the multi-rank control flow of the solver
 * (packed collectives, summable convergence tests, identical branches on every rank): the result must be the one-rank
 * solution with the cost multiplied by `replicas`. */
int vg_comm_create_replicated(vg_comm **out, int replicas, int device);
/* n_ranks communicators for n_ranks host THREADS of this process that drive ONE device, each with its own problem, stream
 * and shard (out: array of n_ranks handles, handle r is rank r; every handle is destroyed by its user).  No RCCL behind it:
 * a collective parks every rank's buffer in a device slot, meets at a host barrier and adds the slots in rank order.  It is a
 * test transport: it runs the solver's real multi-rank data flow -- different shards, ranks without images, the in-place
 * collectives -- on a one-GPU box, which vg_comm_create_replicated (identical shards) cannot.  A rank that fails or does
 * not arrive within 120 s breaks the group: every pending and later collective returns VG_ERR_STATE. */
int vg_comm_create_local(vg_comm **out /* [n_ranks] */, int n_ranks, int device);
int vg_comm_size(const vg_comm *c);
int vg_comm_rank(const vg_comm *c);
/* in-place sum of n doubles at device_buf over all ranks (ncclAllReduce, ncclDouble, ncclSum), enqueued on hip_stream */
int vg_comm_allreduce_sum(vg_comm *c, double *device_buf, int64_t n, void *hip_stream);
void vg_comm_destroy(vg_comm *c);

/* Host-staged alternative (tests on CPU-only boxes, non-RCCL transports): sums host_buf[0..n) over all ranks in place
 * (every rank must end with the same values).  NULL = one GPU. */
typedef int (*vg_allreduce_fn)(double *host_buf, int64_t n, void *user);

typedef struct vg_solve_options {
    int max_num_iterations;             /* 1000   unified_calibration.cpp:46 */
    double function_tolerance;          /* 1e-15  :47 */
    double gradient_tolerance;          /* 1e-15  :48 */
    double parameter_tolerance;         /* 1e-15  :49 */
    double initial_trust_region_radius; /* 1e4    Ceres default (the reference does not set it) */
    double max_trust_region_radius;     /* 1e16 */
    double min_trust_region_radius;     /* 1e-32 */
    double min_relative_decrease;       /* 1e-3 */
    double min_lm_diagonal;             /* 1e-6 */
    double max_lm_diagonal;             /* 1e32 */
    int use_bounds;                     /* 1 */
    int verbose;                        /* minimizer_progress_to_stdout, :51 */
    double soft_l1_scale;               /* 0: no loss function (the global problem, :539-564 pass NULL).  a > 0:
                                           ceres::SoftLOneLoss(a) on every grid residual block, rho(s) =
                                           2 a^2 (sqrt(1 + s / a^2) - 1) of the block's squared norm s -- what the two
                                           initial refinements use (a = 25 :1143, a = 1 :379-401) */
    vg_allreduce_fn allreduce;          /* multi-GPU through host buffers (three calls per iteration) */
    void *allreduce_user;
    vg_comm *comm;                      /* multi-GPU through RCCL: one in-place all-reduce of the device buffer
                                           [summed normal-equation blocks | step scalars] per evaluation and one of the
                                           reduced (Schur) system per linear solve, on the problem's stream.  NULL or a
                                           one-rank communicator = one GPU.  Exclusive with `allreduce`. */
} vg_solve_options;

enum vg_termination {
    VG_TERM_CONVERGENCE_FUNCTION = 0,
    VG_TERM_CONVERGENCE_GRADIENT = 1,
    VG_TERM_CONVERGENCE_PARAMETER = 2,
    VG_TERM_NO_CONVERGENCE = 3, /* max_num_iterations reached */
    VG_TERM_RADIUS_TOO_SMALL = 4,
    VG_TERM_FAILURE = 5
};

typedef struct vg_solve_summary {
    double initial_cost, final_cost; /* 1/2 sum r^2, as Ceres reports it */
    int num_iterations, num_successful_steps, termination;
    double gradient_max_norm, final_radius;
    double total_seconds, evaluate_seconds, schur_seconds, host_seconds; /* host-driven loop: time in evaluations / pose
                                    elimination / host algebra; device-resident loop: evaluate_seconds = all iterations,
                                    host_seconds = set-up (tables, buffers), schur_seconds = 0 */
    int num_global_columns;     /* G */
    int64_t num_pose_blocks;
    char message[160];
} vg_solve_summary;

void vg_solve_options_init(vg_solve_options *o); /* the defaults listed above */
/* Limits: at most 127 global columns (sum of the cameras' K + 6 per global transform: e.g. eight Mei cameras and
 * seven global transforms); any number of pose blocks below 2^28.  Beyond that VG_ERR_INVALID_ARGUMENT. */
int vg_problem_solve(vg_problem *p, const vg_solve_options *options, vg_solve_summary *summary);
/* A solve works in one device block and one pinned host block (index tables, Gram sets, Schur rows ...) which the library
 * keeps for the next solve of the process instead of returning them (allocation was a third of a 10 k-image solve).
 * This frees them; safe at any time between solves. */
void vg_release_cached_memory(void);

/* The per-image pose refinement of estimateInitialGrid (src/calibration/unified_calibration.cpp:1137-1155) for n
 * images at once: n INDEPENDENT problems -- one GenericProjectionJac block with chain {DIRECT} each, intrinsics
 * constant, ceres::SoftLOneLoss(options->soft_l1_scale), every image with its own trust region, step acceptance and
 * convergence tests (one half-wave per image runs its whole solve inside ONE kernel launch).  Host pointers:
 *   board [3 N], corners [n_images][2 N], poses [n_images][6] (in: start, out: result); optional per-image outputs
 *   iterations / final_cost (rho(|r|^2) / 2) / termination (enum vg_termination), each [n_images] or NULL.
 * options NULL = what the reference runs: Ceres' defaults (function / gradient / parameter tolerance 1e-6 / 1e-10 /
 * 1e-8) with max_num_iterations = 500 and SoftLOneLoss(25). */
int vg_refine_poses(int device, void *hip_stream, int model, const double *intrinsics, int n_points, const double *board,
                    int64_t n_images, const double *corners, double *poses, const vg_solve_options *options,
                    int32_t *iterations, double *final_cost, int32_t *termination);

/* the same call with a clock on the launch itself: *kernel_seconds = duration of vg_pose_lm_kernel alone (HIP events on the
 * launch stream), without the uploads and the read-back around it (bench.py section pose_init, tools/bench_calib.py) */
int vg_refine_poses_timed(int device, void *hip_stream, int model, const double *intrinsics, int n_points, const double *board,
                          int64_t n_images, const double *corners, double *poses, const vg_solve_options *options,
                          int32_t *iterations, double *final_cost, int32_t *termination, double *kernel_seconds);

/* The same refinement for a dataset that is RESIDENT in a finalized problem (reference flow: estimateInitialGrid
 * unified_calibration.cpp:1137-1155 refines the poses that addGridResidualBlocks :514-630 then hands to the global problem over
 * the same corners): the dataset's observations and board and the camera's CURRENT intrinsics are read where they lie in HBM;
 * only poses [n_blocks][6] (host, in: start, out: result -- camera-frame board poses, independent of the dataset's chain) and
 * the optional per-image outputs cross the bus, in one copy each way.  kernel_seconds (may be NULL): the launch alone. */
int vg_dataset_refine_poses(vg_problem *p, int dataset_id, double *poses, const vg_solve_options *options, int32_t *iterations,
                            double *final_cost, int32_t *termination, double *kernel_seconds);

/* Host-only helper of the solver, exported so the host logic can be tested without a GPU:
 * solves the symmetric positive definite n x n system A x = b (row-major A, untouched) by Cholesky.
 * Returns VG_ERR_NUMERIC when A is not positive definite. */
int vg_host_cholesky_solve(int n, const double *A, const double *b, double *x);

/* =====================================================================================
 * 5. Calibration-JSON front end -- class GenericCameraCalibration
 *    (include/calibration/unified_calibration.h:91-180): same schema (README.md:36-223), same parse order
 *    (parseTransforms :91-132, parseCameras :134-180, parseData :632-831), same pose initialisation
 *    (initTransforms :431-512, estimateInitialGrid :1066-1158, getInitTransform :311-348, initGlobalTransform
 *    :358-429), same report and image_error_<i>.txt formats.  Grid-reprojection datasets only: "ir_data", and
 *    "images" with pre-extracted corners ("corners_file", same layout as ir_data's "data_file").
 * ===================================================================================== */
typedef struct vg_calibration vg_calibration;
int vg_calibration_create(vg_calibration **out, int device);
void vg_calibration_destroy(vg_calibration *c);
/* addResiduals(infoFileName), :350-356.  Parsing runs on the host; initialising a transform ("init" != "none")
 * refines the poses on the GPU unless the dataset carries the "do_not_solve" flag. */
int vg_calibration_add_file(vg_calibration *c, const char *json_path);
/* compute(), :39-89: assemble the problem, solve (options NULL = the reference's Solver::Options, :42-52). */
int vg_calibration_compute(vg_calibration *c, const vg_solve_options *options, vg_solve_summary *summary);
/* the text compute() prints (:56-83) / the parse-time messages; return the size needed including the NUL */
int64_t vg_calibration_report(vg_calibration *c, char *buf, int64_t size);
int64_t vg_calibration_log(vg_calibration *c, char *buf, int64_t size);
int vg_calibration_num_datasets(const vg_calibration *c);
int vg_calibration_get_intrinsics(vg_calibration *c, const char *camera, double *out, int *count);
int vg_calibration_get_transform(vg_calibration *c, const char *name, int64_t index, double *out6, int64_t *count);
/* The corner list of one image of a dataset as parsed (detectedCornersVec[image], unified_calibration.h:86): *count = number
 * of doubles (2 per board point, 0 for an image without corners); out (may be NULL) receives them. */
int vg_calibration_get_corners(const vg_calibration *c, int dataset, int64_t image, double *out, int64_t *count);
int64_t vg_calibration_num_images(const vg_calibration *c, int dataset);
/* writeImageResidual(dataVec[dataset], path), :1186-1292.  sigma_out: one value per image of the dataset (NULL ok). */
int vg_calibration_write_residuals(vg_calibration *c, int dataset, const char *path, double *sigma_out,
                                   int64_t *outliers_out);
/* Where the front end's wall-clock time went, in seconds, accumulated over every vg_calibration_add_file / _compute /
 * _write_residuals call on this handle (the reference's program is `calib a.json`: parse -> estimateInitialGrid per image ->
 * ceres::Solve -> report; tools/bench_calib.py and bench.py's calib_e2e section print this table). */
typedef struct vg_calibration_timings {
    double read_files_s;        /* reading the JSON files into memory */
    double parse_json_s;        /* JSON text -> values (calibration file, corner / wheel files) */
    double geometric_init_s;    /* 4-corner pose construction (:1066-1135) + getInitTransform (:311-348), host */
    double refine_total_s;      /* the per-image refinements as seen by the host: poses up, kernel, results back (corners: corner_upload_s) */
    double refine_kernel_s;     /* ... the vg_pose_lm_kernel launches alone (HIP events) */
    double global_init_s;       /* initGlobalTransform refinements (:358-429): a batched solve per global transform */
    double assemble_s;          /* compute(): problem assembly, uploads, vg_problem_finalize */
    double solve_s;             /* compute(): vg_problem_solve */
    double readback_s;          /* compute(): parameters back into the maps */
    double residual_eval_s;     /* write_residuals: chain composition + projection of every image (GPU) */
    double residual_format_s;   /* write_residuals: statistics, formatting and writing image_error_<i>.txt */
    int64_t refine_images;      /* images refined by vg_refine_poses */
    int64_t refine_iterations;  /* sum of their LM iterations */
    int64_t refine_max_iterations; /* the slowest image's iteration count */
    int64_t json_bytes;         /* bytes of JSON text parsed */
    int64_t residual_lines;     /* lines written by write_residuals */
    double corner_upload_s;     /* gathering the detected corners into pinned staging and their upload into HBM (once per dataset) */
    int64_t corner_uploads;     /* corner blocks uploaded: one per dataset when every consumer shares it */
    int64_t corner_upload_bytes;
} vg_calibration_timings;
int vg_calibration_get_timings(const vg_calibration *c, vg_calibration_timings *out);
/* Host-only pieces of the pose initialisation, exported so that they can be checked block by block without a GPU (the
 * front end calls the same code):
 *   vg_reconstruct_point  ICamera::reconstructPoint (eucm.h:85-106, ucm.h:81-103, mei.h:90-112): uv[2] -> X[3] = (xn, yn, z);
 *                         VG_ERR_NUMERIC where the reference returns false.
 *   vg_initial_grid_pose  the 4-corner construction of estimateInitialGrid (unified_calibration.cpp:1066-1135), before its
 *                         refinement: board4 [4][3] / corners4 [4][2] at idxUL, idxUR, idxBL, idxBR -> xi6.
 *   vg_init_transform     getInitTransform (:311-348): chain_values [chain_len][6] = current value of every chain member
 *                         (camera side first; the entry at init_index is not read), xi_camera = the camera-frame board pose
 *                         -> the value of member init_index. */
int vg_reconstruct_point(int model, const double *intrinsics, const double *uv, double *X);
int vg_initial_grid_pose(int model, const double *intrinsics, const double *board4, const double *corners4, double *xi6);
int vg_init_transform(int chain_len, const int *status, int init_index, const double *chain_values, const double *xi_camera,
                      double *out6);
/* the same for a member that occurs more than once in the chain: the reference's forward loop stops at the FIRST occurrence
 * (:314-318), its backward loop at the LAST (:327-337); the members in between are read by neither */
int vg_init_transform_range(int chain_len, const int *status, int first_index, int last_index, const double *chain_values,
                            const double *xi_camera, double *out6);
/* transformFromData, include/json.h:36-67: 3 [x,y,theta] / 6 [t,rotvec] / 7 [t,qx,qy,qz,qw] / 12 row-major [R|t] */
int vg_transform_from_values(int n, const double *values, double *out6);

/* =====================================================================================
 * 6. Localization reprojection costs on the same device camera models:
 *      MonoReprojectCost    include/localization/local_cost_functions.h:159-180, src/localization/local_cost_functions.cpp:216-278
 *      SparseReprojectCost  .h:183-208, .cpp:281-391  (with Triangulator::computeRegular, src/reconstruction/triangulator.cpp:145-259)
 *      CameraJacobian       include/projection/jacobian.h:51-119
 *    The reference evaluates these one block at a time inside ceres::Solve; its RANSAC (src/localization/sparse_odom.cpp:
 *    511-606) solves 200 independent few-point problems per frame pair.  A SET keeps the constructor arguments of many
 *    blocks resident in HBM; one evaluation of the whole set is two launches (one lane per block for the transform chain,
 *    one lane per feature for triangulation / projection / the 2 x 6 rows).  The camera is constant (the reference clones
 *    it in the constructor).  xi_base_cam: the base -> camera transform shared by all blocks of the set.
 * ===================================================================================== */
typedef struct vg_reproject_set vg_reproject_set;
/* SparseReprojectCost ctor (.h:185-195) for n_blocks blocks; block b owns points [offsets[b], offsets[b + 1]) of the
 * flat arrays (all HOST pointers): x1 / x2 [total][3] = _xVec1 / _xVec2 (direction vectors in camera frames 1 / 2),
 * p2 [total][2] = _pVec2, size [total] = _sizeVec.  Parameter block [6] (xiOdom), 2 n residuals per block. */
int vg_sparse_reproject_create(vg_reproject_set **out, int device, void *hip_stream, int model, const double *intrinsics,
                               const double *xi_base_cam, int64_t n_blocks, const int64_t *offsets, const double *x1,
                               const double *x2, const double *p2, const double *size);
/* MonoReprojectCost ctor (.h:161-168): five points per block, x1 [n_blocks][5][3], p2 [n_blocks][5][2] (HOST).
 * Parameter blocks [6 (xiOdom), 5 (lengths)], 10 residuals per block. */
int vg_mono_reproject_create(vg_reproject_set **out, int device, void *hip_stream, int model, const double *intrinsics,
                             const double *xi_base_cam, int64_t n_blocks, const double *x1, const double *p2);
int64_t vg_reproject_num_blocks(const vg_reproject_set *s);
int64_t vg_reproject_num_points(const vg_reproject_set *s);
int64_t vg_reproject_block_offset(const vg_reproject_set *s, int64_t block); /* first point of a block; block == n_blocks: total */
/* Evaluate of EVERY block of the set (DEVICE pointers, asynchronous on the set's stream): xi_odom [n_blocks][6];
 * residuals [total][2] (failed projection: the 1e15 pair); jacobian [total][2][6] row-major or NULL: rows 2i / 2i + 1 of
 * block b's [2 n x 6] Jacobian are at point offsets[b] + i.  Reference behaviour kept as written: only the u-row of a point
 * is divided by its size (.cpp:383-389), and the depth part of the Jacobian is exact only for an identity base -> camera
 * rotation (.cpp:355: tBaseCam1 conjugates the odometry rotation once too often). */
int vg_sparse_reproject_evaluate(vg_reproject_set *s, const double *xi_odom, double *residuals, double *jacobian);
/* the same for a MonoReprojectCost set: xi_odom [n_blocks][6], lengths [n_blocks][5]; residuals [n_blocks][10];
 * jac_odom [n_blocks][10][6], jac_lengths [n_blocks][10][5] (row-major, either may be NULL). */
int vg_mono_reproject_evaluate(vg_reproject_set *s, const double *xi_odom, const double *lengths, double *residuals,
                               double *jac_odom, double *jac_lengths);
/* Evaluate of ONE block with ceres::CostFunction::Evaluate's contract (HOST pointers, synchronous): parameters[0] = xiOdom,
 * parameters[1] = the five lengths (mono only); jacobians NULL or an array of 1 (sparse) / 2 (mono) pointers, each NULL or
 * row-major [2 n x block size].  Returns VG_OK where the reference returns true (always). */
int vg_sparse_reproject_block_evaluate(vg_reproject_set *s, int64_t block, double const *const *parameters, double *residuals,
                                       double **jacobians);
int vg_mono_reproject_block_evaluate(vg_reproject_set *s, int64_t block, double const *const *parameters, double *residuals,
                                     double **jacobians);
int vg_reproject_synchronize(vg_reproject_set *s);
void vg_reproject_destroy(vg_reproject_set *s);
/* CameraJacobian (jacobian.h:51-119) for n points: T12 / T23 HOST 6-vectors (T23 NULL: the one-transform constructor),
 * X2 [n][3], grad [n][2] (NULL without dfdxi) DEVICE; outputs DEVICE, either may be NULL: dpdxi [n][2][6] = (dudxi, dvdxi)
 * of CameraJacobian::dpdxi (:75-96), dfdxi [n][6] of ::dfdxi (:99-113).  A point the camera cannot project gives zero rows. */
int vg_camera_jacobian_evaluate(int device, void *hip_stream, int model, const double *intrinsics, const double *T12,
                                const double *T23, int64_t n, const double *X2, const double *grad, double *dpdxi, double *dfdxi);

/* ---- measurement / test hooks.  The library reads no environment variable to change what it computes or how; the A/B
 * switches used by tests/ and tools/ are set here (process-wide, not thread safe): "inline_chain_max_bytes", "gram_force_mfma",
 * "gram_ch1", "gram_no_merge", "max_obs_per_launch", "solver_timing", "solver_host_loop", "solver_device_loop",
 * "solver_no_speculation", "emit_equal_tiles", ... (the list: enum DebugHook, visgeom_amd/csrc/vg_internal.hpp); value 0 restores
 * the default.  The PRODUCTION library (python -m visgeom_amd._build --production, built without VG_DEBUG_HOOKS) has none of them:
 * every switch is its default at compile time and this entry is not exported.  (VG_RCCL_LIBRARY, the path of the RCCL library to
 * bind, is deployment configuration, not a hook.) */
int vg_debug_set(const char *name, long long value);

/* ---- measurement helpers (bench / profiling only): a pure streaming write / copy with the same
 * 16 B-per-lane access pattern as the emit kernel, to calibrate rocprofv3's WRITE_SIZE / FETCH_SIZE
 * and to measure the achievable HBM rate on the box. */
int vg_calib_stream_write(void *hip_stream, double *dst, int64_t n_doubles, double value);
int vg_calib_stream_copy(void *hip_stream, double *dst, const double *src, int64_t n_doubles);
/* what the FP64 vector pipe delivers on this box at the occupancy of the fused Gram kernels (two waves per SIMD): one launch of
 * dependent-FMA chains (eight per lane, `iters` FMAs each) on `hip_stream`; *flops_out = the launch's flop count.  `scratch`:
 * any device buffer of at least 2 x (number of CUs) doubles (never written in practice).  The caller times the launch. */
int vg_calib_fp64_fma(void *hip_stream, double *scratch, int iters, int64_t *flops_out);
/* the bus ceiling of the host-memory route on this box: `reps` blocking hipMemcpy device -> hipHostMalloc memory of `bytes`
 * bytes each (buffers allocated and touched first); seconds_out[reps] receives the duration of every copy. */
int vg_calib_d2h_copies(int device, int64_t bytes, int reps, double *seconds_out);

#ifdef __cplusplus
}
#endif
#endif /* VISGEOM_AMD_H */
