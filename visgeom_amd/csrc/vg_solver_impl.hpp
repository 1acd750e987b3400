// vg_solver_impl.hpp -- Levenberg-Marquardt driver with per-pose Schur elimination (see vg_solver.hpp).
// Host orchestration + the small dense algebra; all O(images) work runs in HIP kernels.  Included at the end
// of vg_solver_tu.hip.  Set-up memory: vg_solver_memory.hpp; priors and odometry-coupled sequences: vg_solver_coupled.hpp.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <vector>

#include "vg_internal.hpp"
#include "vg_gram_valu.hpp"
#include "vg_solver.hpp"
#include "vg_solver_device.hpp"
#include "vg_transf_host.hpp"

using vgi::fail;

#include "vg_solver_memory.hpp"
#include "vg_solver_coupled.hpp"

namespace {

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Cholesky solve of a dense SPD system, row-major; returns false when not positive definite
// Cholesky solve of a small dense SPD system on the host (the reduced system of the host-driven loop: 45 x 45 for the rig, once
// or twice per iteration).  Right-looking: column j is scaled, copied to a contiguous buffer, and every row's remaining entries
// lose L_ij L_kj in ONE pass over contiguous memory -- a loop the compiler vectorises without reassociating anything, where the
// dot-product form (s -= L_rk L_ck over k) is a serial reduction under -ffp-contract=off / no fast-math: 7 us -> ~3 us per
// iteration of the rig.  Every entry still sees its subtractions in increasing column order: the same bits as that form.
// `ws`: n * n + 2 n doubles of workspace (grown here, kept by the caller: no allocation per call).
bool chol_solve(int n, const double *A, const double *b, double *x, std::vector<double> &ws)
{
    if (ws.size() < (size_t)n * n + 2 * (size_t)n) ws.resize((size_t)n * n + 2 * (size_t)n);
    double *L = ws.data(), *col = L + (size_t)n * n, *y = col + n;
    for (int r = 0; r < n; r++)
        for (int c = 0; c <= r; c++) L[(size_t)r * n + c] = A[(size_t)r * n + c];
    for (int j = 0; j < n; j++) {
        const double d = L[(size_t)j * n + j];
        if (!(d > 0.) || !std::isfinite(d)) return false;
        const double ljj = std::sqrt(d);
        L[(size_t)j * n + j] = ljj;
        for (int i = j + 1; i < n; i++) {
            L[(size_t)i * n + j] /= ljj;
            col[i] = L[(size_t)i * n + j];
        }
        for (int i = j + 1; i < n; i++) {
            const double lij = col[i];
            double *row = L + (size_t)i * n;
            for (int k = j + 1; k <= i; k++) row[k] -= lij * col[k];
        }
    }
    for (int r = 0; r < n; r++) {
        double s2 = b[r];
        for (int c = 0; c < r; c++) s2 -= L[(size_t)r * n + c] * y[c];
        y[r] = s2 / L[(size_t)r * n + r];
    }
    for (int r = n - 1; r >= 0; r--) {
        double s2 = y[r];
        for (int c = r + 1; c < n; c++) s2 -= L[(size_t)c * n + r] * x[c];
        x[r] = s2 / L[(size_t)r * n + r];
    }
    return true;
}

bool chol_solve(int n, const double *A, const double *b, double *x)
{
    std::vector<double> ws;
    return chol_solve(n, A, b, x, ws);
}

int launch_dense_gram(hipStream_t st, const double *X, unsigned n_rows, int C, unsigned rows_per_group, unsigned n_groups,
                      double *out)
{
    const int T = (C + 15) / 16;
    const dim3 grid((n_groups + 3) / 4), blk(256);
    if (T == 1) hipLaunchKernelGGL((vg::vg_dense_gram_kernel<1>), grid, blk, 0, st, X, n_rows, C, rows_per_group, n_groups, out);
    else if (T == 2) hipLaunchKernelGGL((vg::vg_dense_gram_kernel<2>), grid, blk, 0, st, X, n_rows, C, rows_per_group, n_groups, out);
    else if (T == 3) hipLaunchKernelGGL((vg::vg_dense_gram_kernel<3>), grid, blk, 0, st, X, n_rows, C, rows_per_group, n_groups, out);
    else if (T == 4) hipLaunchKernelGGL((vg::vg_dense_gram_kernel<4>), grid, blk, 0, st, X, n_rows, C, rows_per_group, n_groups, out);
    else  // wide reduced systems: one wave per (row group, tile pair)
        hipLaunchKernelGGL(vg::vg_dense_gram_pair_kernel, dim3(grid.x, (unsigned)(T * (T + 1) / 2)), blk, 0, st, X, n_rows, C,
                           rows_per_group, n_groups, out);
    VG_HIP(hipGetLastError());
    return VG_OK;
}

}  // namespace

extern "C" {

void vg_release_cached_memory(void)
{
    std::lock_guard<std::mutex> lk(g_arena_cache.m);
    g_arena_cache.drop();
}

void vg_solve_options_init(vg_solve_options *o)
{
    if (!o) return;
    o->max_num_iterations = 1000;   // unified_calibration.cpp:46
    o->function_tolerance = 1e-15;  // :47
    o->gradient_tolerance = 1e-15;  // :48
    o->parameter_tolerance = 1e-15; // :49
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->use_bounds = 1;
    o->verbose = 0;
    o->soft_l1_scale = 0.;
    o->allreduce = nullptr;
    o->allreduce_user = nullptr;
    o->comm = nullptr;
}

int vg_host_cholesky_solve(int n, const double *A, const double *b, double *x)
{
    if (n <= 0 || !A || !b || !x) return fail(VG_ERR_INVALID_ARGUMENT, "bad arguments");
    return chol_solve(n, A, b, x) ? VG_OK : fail(VG_ERR_NUMERIC, "matrix is not positive definite");
}

int vg_problem_solve(vg_problem *p, const vg_solve_options *opt_in, vg_solve_summary *sum)
{
    if (!p) return fail(VG_ERR_INVALID_ARGUMENT, "problem is NULL");
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    vg_solve_options opt;
    if (opt_in) opt = *opt_in;
    else vg_solve_options_init(&opt);
    VG_HIP(hipSetDevice(p->device));
    hipStream_t st = p->stream;
    const double t_start = now_s();
    double t_eval = 0., t_schur = 0., t_host = 0.;
    const bool trace_setup = vgi::debug_hook(vgi::kHookSolverTiming) != 0;  // measurement hook: where the set-up time goes
    double t_mark = t_start;
    auto mark = [&](const char *what) {
        if (!trace_setup) return;
        const double t = now_s();
        std::fprintf(stderr, "[vg_problem_solve] %-28s %8.1f us\n", what, (t - t_mark) * 1e6);
        t_mark = t;
    };

    // ---------------------------------------------------------------- column / pose bookkeeping
    const int n_ds = (int)p->dss.size();
    const int64_t n_params = p->n_params;
    std::vector<int> cam_goff(p->cams.size()), tf_goff(p->tfs.size(), -1);
    std::vector<int64_t> tf_pbase(p->tfs.size(), -1);
    int G = 0;
    for (size_t c = 0; c < p->cams.size(); c++) { cam_goff[c] = G; G += p->cams[c].K; }
    for (size_t t = 0; t < p->tfs.size(); t++)
        if (p->tfs[t].global) { tf_goff[t] = G; G += 6; }
    std::vector<int> pb_goff(p->pblocks.size());
    for (size_t b = 0; b < p->pblocks.size(); b++) { pb_goff[b] = G; G += p->pblocks[b].size; }
    if (G > 127) return fail(VG_ERR_INVALID_ARGUMENT, "more than 127 global columns are not supported");
    int64_t n_poses = 0;
    for (size_t t = 0; t < p->tfs.size(); t++)
        if (!p->tfs[t].global) { tf_pbase[t] = n_poses; n_poses += p->tfs[t].count; }
    if (n_poses > 0x7fffffff / 8) return fail(VG_ERR_INVALID_ARGUMENT, "too many pose blocks");

    std::vector<unsigned char> gfrozen(G, 0), pose_frozen((size_t)n_poses, 0);
    std::vector<long long> gcol_param(G), pose_param((size_t)n_poses);
    // box bounds exist for intrinsics only (eucm.h:228-246, ucm.h:199-215, mei.h:287-313, set at unified_calibration.cpp:
    // 621-627), i.e. for global columns: one pair per column, nothing per pose parameter
    std::vector<double> glo((size_t)G, -std::numeric_limits<double>::infinity()), ghi((size_t)G, std::numeric_limits<double>::infinity());
    for (size_t c = 0; c < p->cams.size(); c++)
        for (int k = 0; k < p->cams[c].K; k++) {
            gfrozen[cam_goff[c] + k] = p->cams[c].constant;
            gcol_param[cam_goff[c] + k] = p->cams[c].offset + k;
            if (opt.use_bounds && !p->cams[c].constant)
                vg_intrinsic_bounds(p->cams[c].model, k, &glo[(size_t)(cam_goff[c] + k)], &ghi[(size_t)(cam_goff[c] + k)]);
        }
    for (size_t b = 0; b < p->pblocks.size(); b++)
        for (int k = 0; k < p->pblocks[b].size; k++) {
            gfrozen[pb_goff[b] + k] = p->pblocks[b].constant;
            gcol_param[pb_goff[b] + k] = p->pblocks[b].offset + k;
        }
    for (size_t t = 0; t < p->tfs.size(); t++) {
        const vgi::Transform &tf = p->tfs[t];
        if (tf.global) {
            for (int k = 0; k < 6; k++) {
                gfrozen[tf_goff[t] + k] = tf.constant;
                gcol_param[tf_goff[t] + k] = tf.offset + k;
            }
        } else {
            for (int64_t i = 0; i < tf.count; i++) {
                pose_frozen[(size_t)(tf_pbase[t] + i)] = tf.constant;
                pose_param[(size_t)(tf_pbase[t] + i)] = tf.offset + 6 * i;
            }
        }
    }

    // sequences coupled by OdometryPrior blocks -> host elimination (pose mode 2); constant elements ("anchor")
    std::vector<CoupledSeq> coupled;
    for (const auto &b : p->odoms) {
        CoupledSeq *cs = nullptr;
        for (auto &c2 : coupled)
            if (c2.tf == b.tf) cs = &c2;
        if (!cs) {
            coupled.emplace_back();
            cs = &coupled.back();
            cs->tf = b.tf;
            cs->pb = tf_pbase[b.tf];
            cs->n = p->tfs[b.tf].count;
            cs->param_off = p->tfs[b.tf].offset;
            cs->frozen.assign((size_t)cs->n, p->tfs[b.tf].constant ? 1 : 0);
        }
        cs->blocks.push_back(b);
    }
    for (const auto &pr : p->priors) {  // TransformationPrior on a sequence = on its element 0
        if (p->tfs[pr.tf].global) continue;
        CoupledSeq *cs = nullptr;
        for (auto &c2 : coupled)
            if (c2.tf == pr.tf) cs = &c2;
        if (!cs) {  // no odometry on this sequence: only element 0 leaves the per-pose GPU path
            coupled.emplace_back();
            cs = &coupled.back();
            cs->tf = pr.tf;
            cs->pb = tf_pbase[pr.tf];
            cs->n = 1;
            cs->param_off = p->tfs[pr.tf].offset;
            cs->frozen.assign(1, p->tfs[pr.tf].constant ? 1 : 0);
        }
        cs->unary.emplace_back((int64_t)0, pr);
    }
    for (const auto &cp : p->const_poses) {
        pose_frozen[(size_t)(tf_pbase[cp.first] + cp.second)] = 1;
        for (auto &c2 : coupled)
            if (c2.tf == cp.first && cp.second < c2.n) c2.frozen[(size_t)cp.second] = 1;
    }
    for (auto &c2 : coupled) {
        std::sort(c2.blocks.begin(), c2.blocks.end(), [](const vgodo::Block &a2, const vgodo::Block &b2) { return a2.i < b2.i; });
        c2.pb_goff = pb_goff;
        for (int64_t i = 0; i < c2.n; i++) pose_frozen[(size_t)(c2.pb + i)] = 2;
        c2.x.resize((size_t)c2.n * 6);
        c2.xc.resize((size_t)c2.n * 6);
    }
    const vg_comm *comm = opt.comm;
    const bool multi_rank = opt.allreduce != nullptr || (comm && comm->n_ranks > 1);
    if (opt.allreduce && comm && comm->n_ranks > 1)
        return fail(VG_ERR_INVALID_ARGUMENT, "give either an RCCL communicator or a host all-reduce callback, not both");
    // Sequences coupled by odometry blocks across ranks: the sequence transform is REPLICATED (every rank holds all of its
    // elements and all of its odometry / prior blocks), only the images that reference it are sharded.  Every rank's GPU then
    // produces the raw V_i, g_i, W_i^T of every element from ITS images, one in-place all-reduce per coupled sequence sums
    // them, and every rank runs the same block-tridiagonal elimination on the same numbers (a few hundred poses; the
    // reference solves them in the same globalProblem, src/calibration/unified_calibration.cpp:53 with the blocks of
    // :661-807).  Only through a device communicator: the host-callback path packs its scalars before the sum.
    if (!coupled.empty() && opt.allreduce)
        return fail(VG_ERR_INVALID_ARGUMENT, "odometry-coupled sequences need a device communicator (vg_solve_options.comm) for a multi-rank solve, not the host all-reduce callback");
    const bool coupled_multi = !coupled.empty() && multi_rank;
    // which loop drives the iterations (see "device loop" below); measurement / A-B hooks (vg_debug_set) force a side
    const bool force_host_loop = vgi::debug_hook(vgi::kHookSolverHostLoop) != 0;
    const bool force_device_loop = vgi::debug_hook(vgi::kHookSolverDeviceLoop) != 0;
    const bool device_loop = coupled.empty() && p->priors.empty() && !opt.allreduce && !force_host_loop && n_ds <= vg::kLmMaxDatasets && (G <= 32 || force_device_loop);
    // Host-driven loop on one rank without host-eliminated sequences: what the host reads every iteration (the Schur Gram,
    // the summed Gram blocks, the step's scalars) is WRITTEN INTO PINNED HOST MEMORY by the kernels that produce it, and
    // the reduced step is read from pinned memory by the back-substitution -- no copy or memset command between two kernels
    // (each one is an engine hand-over of ~10 us on this stack; the rig's iteration has six of them otherwise).
    const bool host_direct = !device_loop && !comm && !opt.allreduce && coupled.empty();

    // per dataset: local -> global column map, pose column offset, pose references
    std::vector<std::vector<int>> lmap(n_ds);
    std::vector<int> inv((size_t)n_ds * (G ? G : 1), -1), Wd(n_ds), pose_off(n_ds, -1), seq_tf(n_ds, -1);
    int Wmax = 1;
    for (int d = 0; d < n_ds; d++) {
        const vgi::Dataset &D = p->dss[d];
        const int K = p->cams[D.camera].K;
        Wd[d] = K + 6 * D.L + 1;
        Wmax = Wd[d] > Wmax ? Wd[d] : Wmax;
        lmap[d].assign(Wd[d] - 1, -1);
        for (int k = 0; k < K; k++) lmap[d][k] = cam_goff[D.camera] + k;
        for (int l = 0; l < D.L; l++) {
            const int t = D.tids[l];
            if (p->tfs[t].global) {
                for (int k = 0; k < 6; k++) lmap[d][K + 6 * l + k] = tf_goff[t] + k;
            } else {
                if (seq_tf[d] >= 0) return fail(VG_ERR_INVALID_ARGUMENT, "a chain may hold at most one sequence transform");
                seq_tf[d] = t;  // exactly one is what the reference requires (unified_calibration.cpp:223-228)
                pose_off[d] = K + 6 * l;
            }
        }
        // the same global transform twice in one chain would need the two column groups merged
        for (size_t a2 = 0; a2 < lmap[d].size(); a2++)
            if (lmap[d][a2] >= 0) {
                if (inv[(size_t)d * G + lmap[d][a2]] >= 0)
                    return fail(VG_ERR_INVALID_ARGUMENT, "a transform appears twice in one chain");
                inv[(size_t)d * G + lmap[d][a2]] = (int)a2;
            }
    }
    std::vector<int> ref_ptr((size_t)n_poses + 1, 0), ref_ds, ref_blk;
    for (int d = 0; d < n_ds; d++)
        if (seq_tf[d] >= 0)
            for (int64_t b = 0; b < p->dss[d].n_blocks; b++) ref_ptr[(size_t)(tf_pbase[seq_tf[d]] + p->dss[d].h_seq[(size_t)b]) + 1]++;
    for (int64_t i = 0; i < n_poses; i++) ref_ptr[(size_t)i + 1] += ref_ptr[(size_t)i];
    ref_ds.resize(ref_ptr.back());
    ref_blk.resize(ref_ptr.back());
    {
        std::vector<int> cur(ref_ptr.begin(), ref_ptr.end() - 1);
        for (int d = 0; d < n_ds; d++)
            if (seq_tf[d] >= 0)
                for (int64_t b = 0; b < p->dss[d].n_blocks; b++) {
                    const size_t i = (size_t)(tf_pbase[seq_tf[d]] + p->dss[d].h_seq[(size_t)b]);
                    ref_ds[cur[i]] = d;
                    ref_blk[cur[i]++] = (int)b;
                }
    }

    mark("host index tables");
    // ---------------------------------------------------------------- device state
    const int C = G + 1;
    const unsigned int n_rows = (unsigned int)(6 * n_poses);
    const unsigned int rows_per_group = 96;  // 16 poses per wave: enough waves to fill the chip at 5 k poses
    const unsigned int n_groups = n_rows ? (n_rows + rows_per_group - 1) / rows_per_group : 0;
    const unsigned int n_slabs = (n_groups + vg::kSlab - 1) / vg::kSlab;
    // the fused rows + Gram launch (vg_schur_rows_gram_kernel): whole poses per workgroup, `sg_batches` batches of
    // `sg_ppw` poses each so that the partials stay in the hundreds and the rows of a workgroup fit 48 KB of LDS
    const int sg_ppw = vg::kSchurThreads / (G + 1);
    int sg_batches = (int)((n_poses + (int64_t)sg_ppw * 512 - 1) / ((int64_t)sg_ppw * 512));
    sg_batches = sg_batches < 1 ? 1 : (sg_batches > vg::kSchurMaxBatches ? vg::kSchurMaxBatches : sg_batches);
    while (sg_batches > 1 && sizeof(double) * (size_t)sg_batches * sg_ppw * (6 * (G + 2) + 28) + 24 * (size_t)vg::kSchurMaxRefs > 64 * 1024) sg_batches--;   // two 1024-thread workgroups fill a CU: 64 KB each is free
    // + V_i | g_i of every pose of the workgroup, gathered once and shared by the pose's lanes (28 doubles per pose)
    const size_t sg_lds = sizeof(double) * ((size_t)sg_batches * sg_ppw * 6 * (G + 2) + (size_t)sg_batches * sg_ppw * 28) + 24 * (size_t)vg::kSchurMaxRefs;
    const int sg_shared = vgi::debug_hook(vgi::kHookSchurPrivateGather) ? 0 : 1;   // A/B hook: every lane gathers for itself
    if (sg_lds > 48 * 1024)
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vg::vg_schur_rows_gram_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sg_lds));
    const unsigned int sg_wgs = (unsigned int)((n_poses + (int64_t)sg_ppw * sg_batches - 1) / ((int64_t)sg_ppw * sg_batches));
    // one device block + one pinned block for the whole solve (SolveArena); sizes: the buffers below, generously rounded
    size_t up_need = 64 * 1024 + (size_t)n_ds * 1024;
    up_need += sizeof(int) * (inv.size() + ref_ptr.size() + ref_ds.size() + ref_blk.size()) + (size_t)n_poses * (1 + sizeof(long long));
    up_need += sizeof(double) * 2 * (size_t)G + sizeof(long long) * (size_t)G + 16 * 256;
    size_t dev_need = up_need + (4u << 20);
    for (int d = 0; d < n_ds; d++) {
        const size_t ww = (size_t)Wd[d] * Wd[d];
        dev_need += 2 * sizeof(double) * ((size_t)p->dss[d].n_blocks * ww + 32) + sizeof(double) * ((size_t)p->dss[d].n_blocks / vg::kSlab + 2) * ww +
                    sizeof(double) * ((size_t)p->dss[d].n_blocks / vg::kValuImagesPerBlock + 2) * ww;  // slab / per-workgroup partial sums
    }
    dev_need += sizeof(double) * (3 * (size_t)n_params + (size_t)n_poses * vg::kPoseRec + (size_t)n_rows * C + ((size_t)(n_groups > sg_wgs ? n_groups : sg_wgs) + n_slabs + 8) * (C * C + 1) +
                                  (size_t)n_poses + 8 * (size_t)C * C);
    const size_t pin_need = up_need + (1u << 20) + sizeof(double) * ((size_t)n_ds * Wmax * Wmax + 4 * (size_t)C * C);
    VG_HIP(hipSetDevice(p->device));
    ArenaScope arena_scope(p->device, dev_need, pin_need, up_need);
    std::vector<DevBuf<double>> gramA_v((size_t)(n_ds ? n_ds : 1)), gramB_v((size_t)(n_ds ? n_ds : 1));  // sized once, never resized
    DevBuf<double> *const gramA = gramA_v.data(), *const gramB = gramB_v.data();
    DevBuf<double> d_sums, d_x, d_xc, d_delta, d_glo, d_ghi, d_rec, d_rows, d_rgroups, d_rgram, d_dg, d_scal;
    DevBuf<vg::SolveDatasetDev> d_dsA, d_dsB;
    DevBuf<int> d_inv, d_ref_ptr, d_ref_ds, d_ref_blk;
    DevBuf<unsigned char> d_pf;
    DevBuf<long long> d_pose_param, d_gcol_param;
    int rc;
    std::vector<vg::SolveDatasetDev> hdsA(n_ds), hdsB(n_ds);
    for (int d = 0; d < n_ds; d++) {
        const size_t n = (size_t)p->dss[d].n_blocks * Wd[d] * Wd[d];
        if ((rc = gramA[d].alloc(n)) != VG_OK || (rc = gramB[d].alloc(n)) != VG_OK) return rc;
        hdsA[d] = {gramA[d].p, Wd[d], pose_off[d]};
        hdsB[d] = {gramB[d].p, Wd[d], pose_off[d]};
    }
#define VG_TRY(e) do { if ((rc = (e)) != VG_OK) return rc; } while (0)
    mark("Gram set allocation");
    VG_TRY(d_dsA.upload(hdsA));
    VG_TRY(d_dsB.upload(hdsB));
    VG_TRY(d_inv.upload(inv));
    VG_TRY(d_ref_ptr.upload(ref_ptr));
    VG_TRY(d_ref_ds.upload(ref_ds));
    VG_TRY(d_ref_blk.upload(ref_blk));
    VG_TRY(d_pf.upload(pose_frozen));
    VG_TRY(d_pose_param.upload(pose_param));
    VG_TRY(d_gcol_param.upload(gcol_param));
    VG_TRY(d_glo.upload(glo));
    VG_TRY(d_ghi.upload(ghi));
    // The frames of a CANDIDATE point are built by the back-substitution kernel that computes the point (VERDICT r3 next #5:
    // one launch less in front of every candidate evaluation): possible when every dataset whose Gram kernel reads frames hangs on
    // a pose this solve eliminates on the device.  vg_debug_set("solver_no_fold_frames", 1): the chain prep launch, as before.
    bool fold_frames = vgi::gram_needs_frames(p) && coupled.empty() && n_poses > 0 && !vgi::debug_hook(vgi::kHookSolverNoFoldFrames);
    for (int d = 0; d < n_ds && fold_frames; d++)
        if (vgi::gram_dataset_needs_frames(p, d) && seq_tf[d] < 0) fold_frames = false;
    DevBuf<vg::PrepDataset> d_fold;
    DevBuf<int> d_fold_gcol;
    if (fold_frames) {
        std::vector<vg::PrepDataset> fold((size_t)n_ds);
        std::vector<int> fold_gcol((size_t)n_ds * vg::kMaxChain, -1);
        for (int d = 0; d < n_ds; d++) {
            const vgi::Dataset &D = p->dss[d];
            vg::PrepDataset &pd = fold[(size_t)d];
            pd.chain = D.chain;
            pd.seq_index = D.seq_identity ? nullptr : D.d_seq;
            pd.frames = D.d_frames;
            pd.first = 0;
            pd.count = vgi::gram_dataset_needs_frames(p, d) ? D.n_blocks : 0;
            pd.frame_stride_d = D.frame_stride;
            for (int l = 0; l < D.L; l++)
                if (p->tfs[D.tids[l]].global) fold_gcol[(size_t)d * vg::kMaxChain + l] = tf_goff[D.tids[l]];
        }
        VG_TRY(d_fold.upload(fold));
        VG_TRY(d_fold_gcol.upload(fold_gcol));
    }
    mark("uploads");
    VG_TRY(d_sums.alloc((size_t)n_ds * Wmax * Wmax + 5));
    VG_TRY(d_x.alloc((size_t)n_params));
    VG_TRY(d_xc.alloc((size_t)n_params));
    VG_TRY(d_delta.alloc((size_t)n_params));
    VG_TRY(d_rec.alloc((size_t)n_poses * vg::kPoseRec));
    VG_TRY(d_rows.alloc((size_t)n_rows * C));
    VG_TRY(d_rgroups.alloc((size_t)(n_groups > sg_wgs ? n_groups : sg_wgs) * (C * C + 1)));  // fused rows + Gram: C * C + 1 per workgroup
    // [Gram of the pose rows (C x C) | number of pose blocks that were not positive definite]: ONE buffer, so that the
    // count is summed over ranks by the same all-reduce and every rank takes the same accept / reject branch
    VG_TRY(d_rgram.alloc((size_t)C * C + 1));
    double *const d_bad = d_rgram.p + (size_t)C * C;
    VG_TRY(d_dg.alloc((size_t)(G ? G : 1)));
    const unsigned int n_bs_groups = (unsigned int)((n_poses + vg::kBsPosesPerBlock - 1) / vg::kBsPosesPerBlock);
    // d_sums = [per-dataset summed Gram blocks (n_ds x Wmax^2) | scalar sums of the step (5)]: everything that is SUMMED
    // over ranks, contiguous, so that one evaluation ends with ONE in-place RCCL all-reduce of this buffer on the
    // problem's stream and one D2H.  d_small = [max |g_pose| (bit pattern) 1 | current global values G].
    const size_t n_sums = (size_t)n_ds * Wmax * Wmax, n_pack = n_sums + 5;
    DevBuf<double> d_small;
    struct { double *p; } d_scal_sum{nullptr}, d_xg{nullptr};
    struct { unsigned long long *p; } d_gmax{nullptr};
    VG_TRY(d_scal.alloc((size_t)n_bs_groups * 5));
    VG_TRY(d_small.alloc((size_t)1 + (size_t)(G ? G : 1)));
    // what is cleared / copied before the first evaluation: collected here, done by ONE launch at the head of the loop that runs
    vg::SolverInitArgs init;
    std::memset(&init.h0, 0, sizeof init.h0);
    init.add_zero(d_small.p, 1 + (size_t)(G ? G : 1));
    init.add_zero(d_sums.p, n_pack);
    d_gmax.p = reinterpret_cast<unsigned long long *>(d_small.p);
    d_xg.p = d_small.p + 1;
    init.add_zero(d_delta.p, (size_t)(n_params ? n_params : 1));
    init.src = p->d_params;
    init.dst0 = d_x.p;
    init.n_copy = (unsigned long long)n_params;
    auto launch_init = [&]() -> int {
        unsigned long long n_max = init.n_copy;
        for (int k = 0; k < vg::SolverInitArgs::kZero; k++) n_max = init.n_zero[k] > n_max ? init.n_zero[k] : n_max;
        const unsigned int grid = (unsigned int)std::min<unsigned long long>(std::max<unsigned long long>((n_max + 255) / 256, 1ull), 1024ull);
        hipLaunchKernelGGL(vg::vg_solver_init_kernel, dim3(grid), dim3(256), 0, st, init);
        VG_HIP(hipGetLastError());
        return VG_OK;
    };

    std::vector<double> h_sums((size_t)n_ds * Wmax * Wmax + 5), h_rgram((size_t)C * C + 1);
    std::vector<double> U((size_t)G * G), gg(G), Uc((size_t)G * G), ggc(G), S((size_t)G * G), rhs(G), dg(G), h_xg(G);
    PinnedBuf pin_sums, pin_rgram, pin_small;
    long long n_bad_pose_blocks = 0;
    // pin_small: [dg (G) | gmax (1) | xg (G)]
    VG_TRY(pin_sums.alloc(h_sums.size()));
    VG_TRY(pin_rgram.alloc(h_rgram.size()));
    VG_TRY(pin_small.alloc((size_t)2 * G + 2));
    // where the sum kernels deliver [summed Gram blocks | 5 step scalars]: the device buffer (all-reduced / read by the accept
    // kernel), or straight into the pinned block the host reads
    double *const sums_out = host_direct ? pin_sums.p : d_sums.p;
    if (host_direct) {
        std::memset(pin_sums.p, 0, sizeof(double) * h_sums.size());
        std::memset(pin_small.p, 0, sizeof(double) * ((size_t)2 * G + 2));
    }
    d_scal_sum.p = sums_out + n_sums;
    // The host-driven loop on one rank does not wait through the runtime for the two read-backs of an iteration: a sequence
    // number lands in pinned memory and the host spins on it -- hipStreamSynchronize costs ~3.5 us more per round trip
    // (tools/exp/host_wait.hip).  The strided sum of the pose elimination (133 workgroups at G = 45) stores it when its last
    // workgroup is done (vg::HostSignal); behind an evaluation, whose last launch has 800 workgroups at the rig's size (an atomic
    // each cost more than the wait saves), a one-thread kernel does.  Rig, same box: 0.163-0.170 -> 0.157-0.158 ms per iteration.
    // vg_debug_set("solver_event_wait", 1): the runtime's wait (A/B).
    const bool host_spin = host_direct && !vgi::debug_hook(vgi::kHookSolverEventWait);
    PinnedBuf pin_seq;
    DevBuf<unsigned int> d_sigcnt;
    unsigned long long seq_issued[2] = {0ull, 0ull};
    if (host_spin) {
        VG_TRY(pin_seq.alloc(2));
        VG_TRY(d_sigcnt.alloc(2));
        init.add_zero(reinterpret_cast<double *>(d_sigcnt.p), 1);   // two 32-bit counters
        std::memset(pin_seq.p, 0, sizeof(double) * 2);
    }
    auto host_signal = [&](int which) {   // 0: evaluation, 1: pose elimination
        vg::HostSignal h;
        h.counter = d_sigcnt.p + which;
        h.host_seq = reinterpret_cast<unsigned long long *>(pin_seq.p) + which;
        h.seq = ++seq_issued[which];
        return h;
    };
    auto host_wait = [&](int which) -> int {
        volatile unsigned long long *w = reinterpret_cast<volatile unsigned long long *>(pin_seq.p) + which;
        const double t_spin = now_s();
        unsigned long spins = 0;
        while (*w != seq_issued[which]) {
            if ((++spins & 0xfffff) == 0 && now_s() - t_spin > 30.) {   // the device is gone or a launch failed: do not hang
                VG_HIP(hipStreamSynchronize(st));
                if (*w != seq_issued[which]) return fail(VG_ERR_STATE, "a sum kernel of an LM iteration never reported");
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        return VG_OK;
    };

    mark("scratch + pinned allocation");
    // several datasets: their fixed-order sums run as ONE slab launch and ONE final launch (descriptor tables for
    // the two alternating Gram sets); a single dataset keeps the plain kernels
    std::vector<DevBuf<double>> sum_partials(n_ds);
    DevBuf<vg::SumDataset> d_sumA, d_sumB;
    unsigned int sum_slab_blocks = 0, sum_final_blocks = 0;
    if (n_ds > 1) {
        std::vector<vg::SumDataset> ta, tb;
        for (int d = 0; d < n_ds; d++) {
            const unsigned int n = (unsigned int)p->dss[d].n_blocks;
            if (!n) continue;   // its slot of d_sums stays zero (cleared below)
            vg::SumDataset sd;
            sd.n_items = n;
            sd.n_slabs = (n + vg::kSlab - 1) / vg::kSlab;
            sd.entries = Wd[d] * Wd[d];
            VG_TRY(sum_partials[d].alloc((size_t)sd.n_slabs * sd.entries));
            sd.partials = sum_partials[d].p;
            sd.out = sums_out + (size_t)d * Wmax * Wmax;
            sd.first_slab_block = sum_slab_blocks;
            sd.first_final_block = sum_final_blocks;
            sum_slab_blocks += sd.n_slabs;
            sum_final_blocks += (unsigned int)((sd.entries + 3) / 4);
            sd.gram = gramA[d].p;
            ta.push_back(sd);
            sd.gram = gramB[d].p;
            tb.push_back(sd);
        }
        VG_TRY(d_sumA.upload(ta));
        VG_TRY(d_sumB.upload(tb));
    }
    // several datasets, all of them on the merged vector-pipe launch and no loss function: the launch leaves per-workgroup
    // partial sums and ONE launch adds them -- the slab pass, which reads every Gram block again, is not needed
    const bool use_partials = n_ds > 1 && !(opt.soft_l1_scale > 0.) && vgi::gram_merge_covers_all(p);
    std::vector<DevBuf<double>> wg_partials((size_t)n_ds);
    std::vector<double *> wg_partials_ptr((size_t)n_ds, nullptr);
    DevBuf<vg::PartialSumDataset> d_psum;
    unsigned int psum_blocks = 0;
    int n_psum = 0;
    if (use_partials) {
        std::vector<vg::PartialSumDataset> tab;
        for (int d = 0; d < n_ds; d++) {
            if (!p->dss[d].n_blocks) continue;  // its slot of d_sums stays zero
            vg::PartialSumDataset pd;
            pd.n_wg = (unsigned int)((p->dss[d].n_blocks + vg::kValuImagesPerBlock - 1) / vg::kValuImagesPerBlock);
            pd.W = Wd[d];
            const int E = Wd[d] * (Wd[d] + 1) / 2;
            VG_TRY(wg_partials[(size_t)d].alloc((size_t)E * pd.n_wg));
            wg_partials_ptr[(size_t)d] = wg_partials[(size_t)d].p;
            pd.partials = wg_partials[(size_t)d].p;
            pd.out = sums_out + (size_t)d * Wmax * Wmax;
            pd.first_block = psum_blocks;
            psum_blocks += (unsigned int)E;
            tab.push_back(pd);
        }
        n_psum = (int)tab.size();
        VG_TRY(d_psum.upload(tab));
    }
    // queue the evaluation of the Gram matrices at a device parameter buffer into gram set `set`, their fixed-order sums
    // into d_sums and the ONE collective of an evaluation (no host synchronisation)
    // `step_scalars`: the five scalar sums of the step that led to x_dev (+ its max |g_pose| to `gmax_out`) are wanted with this
    // evaluation: added by two more workgroups of the partial-sum launch when there is one, by vg_step_scalars_kernel otherwise
    auto enqueue_evaluate = [&](const double *x_dev, DevBuf<double> *set, bool frames_ready = false, bool step_scalars = false,
                                unsigned long long *gmax_out = nullptr) -> int {
        int r;
        if (step_scalars && !use_partials) {
            hipLaunchKernelGGL(vg::vg_step_scalars_kernel, dim3(2), dim3(256), 0, st, (const double *)d_scal.p, n_bs_groups, d_scal_sum.p,
                               (const unsigned long long *)d_gmax.p, gmax_out);
            VG_HIP(hipGetLastError());
        }
        if (vgi::gram_needs_frames(p) && !frames_ready && (r = vgi::prepare_at(p, x_dev)) != VG_OK) return r;
        // several datasets: the ones the vector-pipe kernel takes share one launch
        std::vector<char> merged((size_t)n_ds, 0);
        if (n_ds > 1) {
            std::vector<double *> gp((size_t)n_ds);
            for (int d = 0; d < n_ds; d++) gp[(size_t)d] = set[d].p;
            if ((r = vgi::gram_fused_merged_at(p, x_dev, gp.data(), merged, use_partials ? wg_partials_ptr.data() : nullptr)) != VG_OK) return r;
        }
        for (int d = 0; d < n_ds; d++) {
            double *sum_d = sums_out + (size_t)d * Wmax * Wmax;
            const bool robust = opt.soft_l1_scale > 0. && p->dss[d].n_blocks;
            // single dataset, no loss function: Gram blocks and their sum in two launches
            const bool fused_sum = !sum_slab_blocks && !robust;
            if (!merged[(size_t)d] && (r = vgi::gram_fused_at(p, d, x_dev, set[d].p, fused_sum ? sum_d : nullptr)) != VG_OK) return r;
            if (robust) {
                // robustified blocks: J'^T J' = rho' J^T J, J'^T r' = rho' J^T r, cost term rho(s)   (Ceres' Corrector
                // with rho'' < 0, always the case for SoftLOne) -- re-weight the Gram blocks in place, nothing
                // downstream changes
                hipLaunchKernelGGL(vg::vg_gram_soft_l1_kernel, dim3((unsigned)p->dss[d].n_blocks), dim3(64), 0, st, set[d].p,
                                   Wd[d] * Wd[d], opt.soft_l1_scale * opt.soft_l1_scale);
                VG_HIP(hipGetLastError());
            }
            if (!sum_slab_blocks && !fused_sum && (r = vgi::gram_sum_into(p, d, set[d].p, sum_d)) != VG_OK) return r;
        }
        if (use_partials) {
            vg::StepScalarsArgs ssa;
            if (step_scalars) {
                ssa.in = d_scal.p;
                ssa.n_items = n_bs_groups;
                ssa.out = d_scal_sum.p;
                ssa.gmax_bits = d_gmax.p;
                ssa.gmax_out = gmax_out;
                ssa.first_block = psum_blocks;
            }
            hipLaunchKernelGGL(vg::vg_gram_partials_sum_multi_kernel, dim3(psum_blocks + (step_scalars ? 2u : 0u)), dim3(256), 0, st,
                               (const vg::PartialSumDataset *)d_psum.p, n_psum, ssa);
            VG_HIP(hipGetLastError());
        } else if (sum_slab_blocks) {
            const vg::SumDataset *tab = set == gramA ? d_sumA.p : d_sumB.p;
            int n_tab = 0;
            for (int d = 0; d < n_ds; d++) n_tab += p->dss[d].n_blocks ? 1 : 0;
            hipLaunchKernelGGL(vg::vg_gram_slab_sum_multi_kernel, dim3(sum_slab_blocks), dim3(256), 0, st, tab, n_tab);
            VG_HIP(hipGetLastError());
            hipLaunchKernelGGL(vg::vg_gram_final_sum_multi_kernel, dim3(sum_final_blocks), dim3(256), 0, st, tab, n_tab);
            VG_HIP(hipGetLastError());
        }
        // the ONE collective of an evaluation: [summed Gram blocks | scalar sums of the step], device buffer, in place.
        // In place means that after the first collective every slot holds a cross-rank total: whatever this rank does
        // not rewrite before the next one (the block of a dataset without images here, the scalar tail of a rank
        // without poses) has to be cleared, or that total is added in again.
        if (comm && comm->n_ranks > 1) {
            for (int d = 0; d < n_ds; d++)
                if (!p->dss[d].n_blocks) VG_HIP(hipMemsetAsync(d_sums.p + (size_t)d * Wmax * Wmax, 0, sizeof(double) * Wmax * Wmax, st));
            if (!n_bs_groups) VG_HIP(hipMemsetAsync(d_sums.p + n_sums, 0, sizeof(double) * 5, st));
        }
        if (host_direct) return VG_OK;  // one rank, the sums are already where the host reads them
        return vgc::allreduce_sum(comm, d_sums.p, n_pack, st);
    };
    // evaluate at a device parameter buffer and assemble U / gg / cost on the host
    auto evaluate = [&](const double *x_dev, DevBuf<double> *set, std::vector<double> &Uo, std::vector<double> &go,
                        double &cost2, bool frames_ready = false, bool step_scalars = false, unsigned long long *gmax_out = nullptr) -> int {
        const double t0 = now_s();
        int r;
        const bool spin = host_spin;   // a one-thread kernel behind the evaluation reports (see host_spin)
        r = enqueue_evaluate(x_dev, set, frames_ready, step_scalars, gmax_out);
        if (r != VG_OK) return r;
        if (spin) {
            hipLaunchKernelGGL(vg::vg_host_flag_kernel, dim3(1), dim3(1), 0, st, host_signal(0));
            VG_HIP(hipGetLastError());
        }
        if (!host_direct) VG_HIP(hipMemcpyAsync(pin_sums.p, d_sums.p, sizeof(double) * h_sums.size(), hipMemcpyDeviceToHost, st));
        if (spin) {
            if ((r = host_wait(0)) != VG_OK) return r;
        } else {
            VG_HIP(hipStreamSynchronize(st));
        }
        // (read where the device wrote it: no staging copy of the 17 KB a rig's blocks are)
        std::fill(Uo.begin(), Uo.end(), 0.);
        std::fill(go.begin(), go.end(), 0.);
        cost2 = 0.;
        for (int d = 0; d < n_ds; d++) {
            const int W = Wd[d];
            const double *Sd = pin_sums.p + (size_t)d * Wmax * Wmax;
            // (the lower triangle only, mirrored: the blocks are symmetric bit for bit, and what the device wrote into pinned
            //  memory is a cache miss per line on the host -- reading the 17 KB of a rig's blocks was most of this loop's time)
            const double *last = Sd + (size_t)(W - 1) * W;   // the residual row: J^T r and r^T r in one contiguous run
            for (int a2 = 0; a2 < W - 1; a2++) {
                const int ga = lmap[d][a2];
                if (ga < 0) continue;
                for (int b2 = 0; b2 <= a2; b2++) {
                    const int gb = lmap[d][b2];
                    if (gb < 0) continue;
                    const double v = Sd[a2 * W + b2];
                    Uo[(size_t)ga * G + gb] += v;
                    if (a2 != b2) Uo[(size_t)gb * G + ga] += v;
                }
                go[ga] += last[a2];
            }
            cost2 += last[W - 1];
        }
        t_eval += now_s() - t0;
        return VG_OK;
    };
    // sum a packed host buffer over ranks (multi-GPU); identity on one GPU
    auto allreduce = [&](std::vector<double> &buf) -> int {
        if (!opt.allreduce) return VG_OK;
        return opt.allreduce(buf.data(), (int64_t)buf.size(), opt.allreduce_user) == 0
                   ? VG_OK
                   : fail(VG_ERR_STATE, "allreduce callback failed");
    };

    // TransformationPrior blocks live on the host: r = A [R e_t; R e_r], e = prior^-1 o xi, Jacobian = A
    // (calib_cost_functions.cpp:214-228).  Added AFTER the all-reduce, identically on every rank.
    auto add_priors = [&](const std::vector<double> &xg_vals, std::vector<double> &Uo, std::vector<double> &go, double &c2) {
        for (const vgi::Prior &pr : p->priors) {
            if (!p->tfs[pr.tf].global) continue;  // element 0 of a sequence: handled with the poses (CoupledSeq::unary)
            const int g0 = tf_goff[pr.tf];
            double r[6];
            prior_residual(pr, &xg_vals[g0], r);
            for (int a2 = 0; a2 < 6; a2++) {
                for (int b2 = 0; b2 < 6; b2++) {
                    double h = 0.;
                    for (int k = 0; k < 6; k++) h += pr.A[6 * k + a2] * pr.A[6 * k + b2];
                    Uo[(size_t)(g0 + a2) * G + g0 + b2] += h;
                }
                double gsum = 0.;
                for (int k = 0; k < 6; k++) gsum += pr.A[6 * k + a2] * r[k];
                go[g0 + a2] += gsum;
            }
            for (int k = 0; k < 6; k++) c2 += r[k] * r[k];
        }
    };
    // ================================================================== device loop
    // Problems made of grid blocks only (no priors, no odometry, no host-staged all-reduce): the reduced solve and the
    // step acceptance run in two one-workgroup kernels, the trust-region state lives on the device, and an iteration
    // is a fixed sequence of eight launches; the host only reads the state the accept kernel publishes in pinned memory
    // (the first version synchronised three times per iteration and copied G doubles one by one).
    // Measured (tools/exp/solve_probe.py, tools/prof_solve.py, one MI355X): 10 k EUCM images 0.110 ms per iteration
    // against 0.21 for the host-driven loop, Mei 0.118; the 45-column rig 0.338 against 0.293 -- there the one-workgroup
    // factorisation of the reduced system costs more than the host's round trip, so wide systems keep the host loop.
    // vg_debug_set("solver_host_loop" / "solver_device_loop") force a side.
    if (device_loop) {
        DevBuf<vg::LmState> d_state;
        DevBuf<double> d_U, d_gvec, d_S, d_xcur;
        DevBuf<int> d_Wd;
        DevBuf<unsigned char> d_gfrozen;
        VG_TRY(d_state.alloc(1));
        VG_TRY(d_U.alloc((size_t)2 * G * G));
        VG_TRY(d_gvec.alloc((size_t)2 * G));
        VG_TRY(d_S.alloc((size_t)G * G));
        VG_TRY(d_xcur.alloc((size_t)G));
        VG_TRY(d_Wd.upload(Wd));
        VG_TRY(d_gfrozen.upload(gfrozen));
        vg::LmState &h0 = init.h0;
        h0.radius = opt.initial_trust_region_radius;
        h0.decrease_factor = 2.;
        h0.mu = 1. / h0.radius;
        h0.term = VG_TERM_NO_CONVERGENCE;
        init.state = d_state.p;
        init.add_zero(d_rgram.p, h_rgram.size());  // also the bad-pose counter behind it
        init.dst1 = d_xc.p;
        vg::LmState final_state;

        vg::LmAcceptArgs aa;
        aa.st = d_state.p;
        aa.U = d_U.p;
        aa.gg = d_gvec.p;
        aa.sums = d_sums.p;
        aa.inv = d_inv.p;
        aa.Wd = d_Wd.p;
        aa.dg = d_dg.p;
        aa.gmax_bits = d_gmax.p;
        aa.bad = d_bad;
        aa.xcur = d_xcur.p;
        aa.x = d_x.p;
        aa.gcol_param = d_gcol_param.p;
        aa.lo = d_glo.p;
        aa.hi = d_ghi.p;
        aa.gfrozen = d_gfrozen.p;
        aa.n_ds = n_ds;
        aa.Wmax = Wmax;
        aa.G = G;
        aa.init = 1;
        aa.multi_rank = multi_rank ? 1 : 0;
        aa.scal_partials = (n_bs_groups && !multi_rank) ? d_scal.p : nullptr;
        aa.n_scal = n_bs_groups;
        size_t accept_lds = sizeof(double) * ((size_t)n_ds * Wmax * Wmax + ((size_t)n_ds * G + 1) / 2 + 1);
        if (accept_lds > 48 * 1024) accept_lds = 0;  // many datasets: read from global memory
        aa.lds_doubles = accept_lds / sizeof(double);
        aa.dmin = opt.min_lm_diagonal;
        aa.dmax = opt.max_lm_diagonal;
        aa.ftol = opt.function_tolerance;
        aa.gtol = opt.gradient_tolerance;
        aa.ptol = opt.parameter_tolerance;
        aa.min_rel_decrease = opt.min_relative_decrease;
        aa.max_radius = opt.max_trust_region_radius;
        aa.min_radius = opt.min_trust_region_radius;
        vg::LmSolveArgs ra;
        ra.st = d_state.p;
        ra.U = d_U.p;
        ra.gg = d_gvec.p;
        ra.rgram = d_rgram.p;
        ra.lo = d_glo.p;
        ra.hi = d_ghi.p;
        ra.gfrozen = d_gfrozen.p;
        ra.xcur = d_xcur.p;
        ra.dg = d_dg.p;
        ra.S = d_S.p;
        ra.G = G;
        ra.use_bounds = opt.use_bounds;
        ra.dmin = opt.min_lm_diagonal;
        ra.dmax = opt.max_lm_diagonal;
        const bool s_in_lds = sizeof(double) * (2 * (size_t)G * G + 4 * (size_t)G + 2) <= 150 * 1024;
        if (s_in_lds) ra.S = nullptr;
        const size_t solve_lds = sizeof(double) * ((s_in_lds ? 2 : 1) * (size_t)G * G + 4 * (size_t)G + 2);
        if (solve_lds > 64 * 1024)  // up to 127 global columns: 133 KB of the CU's 160 KB
            VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vg::vg_lm_reduced_solve_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_lds));
        if (G <= vg::kEntrySolveMaxG && sizeof(double) * vg::lm_entry_solve_lds_doubles(G) > 48 * 1024)   // 51 KB at G = 63
            VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vg::vg_lm_reduced_solve_entries_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * vg::lm_entry_solve_lds_doubles(G))));

        aa.gate_expect = -1;
        ra.gate_expect = -1;
        // An iteration is a fixed sequence of launches whose buffers depend only on the PARITY of the number of accepted
        // steps so far (which Gram set / parameter buffer is "current"); the device keeps that parity in LmState::gate.
        // So the host queues iteration k + 1 for the parity an acceptance of step k would give BEFORE it knows the
        // outcome of step k -- every kernel of a queued iteration returns at once if the gate says otherwise (a
        // rejected step: the same parity is queued again; convergence: gate = -1) -- and only then waits for the state
        // of iteration k.  The GPU always has the next iteration in its queue: no launch latency, no idle time behind
        // the host's read-back.  Robust (SoftLOne) evaluations re-weight the Gram set in place with an ungated kernel, so
        // those solves queue one iteration at a time.  Several ranks speculate too: every rank holds the same state, so every
        // rank queues the same launches and the same collectives; a collective of an iteration that skips itself is NOT
        // skipped -- it runs on every rank, on buffers nobody reads (whatever a real iteration reads it has rewritten or
        // cleared before its own collective).
        // MEASURED (tools/exp/solve_probe.py, 10 k images, state published by the accept kernel itself): EUCM 0.111 vs
        // 0.115 ms per iteration, Mei 0.119 vs 0.126 -- the iteration is bound by its eight dependent launches on the GPU.
        // Replaying the gated iteration as a hipGraph (one per parity) was slower than queueing its launches: 0.120 /
        // 0.126 ms (profiles/NOTES.md).  vg_debug_set("solver_no_speculation", 1) queues one iteration at a time.
        const bool speculate = opt.soft_l1_scale <= 0. && vgi::debug_hook(vgi::kHookSolverNoSpeculation) != 1;
        DevBuf<double> *gset[2] = {gramA, gramB};
        vg::SolveDatasetDev *dset[2] = {d_dsA.p, d_dsB.p};
        double *xbuf[2] = {d_x.p, d_xc.p};
        constexpr int kSlots = 4;
        struct Slots {
            vg::LmState *p = nullptr;
            volatile unsigned long long *seq = nullptr;   // pinned, behind the states: what the accept kernel of a slot wrote last
            unsigned long long expect[kSlots] = {};
            bool owned = false;
            hipEvent_t ev[kSlots] = {};
            ~Slots()
            {
                if (p && owned) (void)hipHostFree(p);
                for (auto e : ev)
                    if (e) (void)hipEventDestroy(e);
            }
        } slots;
        const size_t slots_bytes = sizeof(vg::LmState) * kSlots + sizeof(unsigned long long) * kSlots;
        if (t_arena) slots.p = static_cast<vg::LmState *>(t_arena->pin_alloc(slots_bytes));
        if (!slots.p) {
            VG_HIP(hipHostMalloc(reinterpret_cast<void **>(&slots.p), slots_bytes, hipHostMallocDefault));
            slots.owned = true;
        }
        slots.seq = reinterpret_cast<volatile unsigned long long *>(slots.p + kSlots);
        for (int k = 0; k < kSlots; k++) slots.seq[k] = 0ull;
        // The host learns the outcome of an iteration by SPINNING on the slot's sequence word, which the accept kernel stores
        // (system-scope release) behind the state -- not from an event recorded behind the kernel: the event's marker packet kept
        // the next iteration's first kernel waiting 5-6 us after every accept (rocprofv3 trace, tools/exp/trace_gaps.py).
        // vg_debug_set("solver_event_wait", 1) restores the event (A/B).
        const bool spin_wait = !vgi::debug_hook(vgi::kHookSolverEventWait);
        if (!spin_wait)
            for (auto &e : slots.ev) VG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        unsigned long long seq_counter = 0ull;
        int n_queued = 0;
        // the accept kernel writes its state into pinned slot `slot` itself; the sequence word (or the event) tells the host when
        auto next_slot = [&]() { return n_queued++ % kSlots; };
        auto arm_slot = [&](int slot, vg::LmAcceptArgs &args) {
            args.host_state = slots.p + slot;
            if (spin_wait) {
                slots.expect[slot] = ++seq_counter;
                args.host_seq = const_cast<unsigned long long *>(slots.seq + slot);
                args.seq = slots.expect[slot];
            }
        };
        auto queue_state = [&](int slot) -> int {
            if (!spin_wait) VG_HIP(hipEventRecord(slots.ev[slot], st));
            return VG_OK;
        };
        // queue one LM iteration for parity `par` (current point = set / buffer `par`, candidate = the other one)
        auto queue_iteration = [&](int par, bool gated, int &slot) -> int {
            const int *gate = gated ? &d_state.p->gate : nullptr;
            vg::SchurArgs sa;
            sa.ds = dset[par];
            sa.inv = d_inv.p;
            sa.ref_ptr = d_ref_ptr.p;
            sa.ref_ds = d_ref_ds.p;
            sa.ref_blk = d_ref_blk.p;
            sa.pose_frozen = d_pf.p;
            sa.n_ds = n_ds;
            sa.G = G;
            sa.n_poses = (int)n_poses;
            sa.mu = 0.;
            sa.mu_dev = &d_state.p->mu;
            sa.dmin = opt.min_lm_diagonal;
            sa.dmax = opt.max_lm_diagonal;
            sa.rec = d_rec.p;
            sa.rows = d_rows.p;
            sa.bad = d_bad;
            sa.gate = gate;
            sa.gate_expect = par;
            if (n_poses) {
                // rows of every pose + the Gram of the rows, one launch; then ONE fixed-order sum over the workgroups
                hipLaunchKernelGGL(vg::vg_schur_rows_gram_kernel, dim3(sg_wgs), dim3(vg::kSchurThreads * sg_batches), sg_lds, st, sa, sg_ppw, sg_batches, d_rgroups.p, sg_shared);
                VG_HIP(hipGetLastError());
                vg::launch_strided_sum(st, d_rgroups.p, sg_wgs, C * C + 1, d_rgram.p);  // the Gram and the count of bad pose blocks
                VG_HIP(hipGetLastError());
            } else if (multi_rank) {
                // a rank without poses still joins the sum: the buffer holds the cross-rank total of the previous iteration
                VG_HIP(hipMemsetAsync(d_rgram.p, 0, sizeof(double) * h_rgram.size(), st));
            }
            VG_TRY(vgc::allreduce_sum(comm, d_rgram.p, h_rgram.size(), st));  // Schur complement of the poses of all ranks
            vg::LmSolveArgs r2 = ra;
            r2.gate_expect = gated ? par : -1;
            // every back-substitution workgroup solves the reduced system itself -- while there are few enough of them: the
            // redundant solves are SIMD time (~1 500 instructions per wave and workgroup), at 100 k poses (3 125 workgroups) they
            // made the launch 82 us where a one-workgroup solve launch + the plain back-substitution take 30
            const long long fold_max_groups = vgi::debug_hook(vgi::kHookSolverFoldMaxGroups) ? vgi::debug_hook(vgi::kHookSolverFoldMaxGroups) : vg::kFoldMaxGroups;
            const bool fold_solve = G > 0 && G <= vg::kFoldMaxG && (long long)n_bs_groups <= fold_max_groups;
            if (!fold_solve) {
                if (G <= vg::kEntrySolveMaxG)
                    hipLaunchKernelGGL(vg::vg_lm_reduced_solve_entries_kernel, dim3(1), dim3(vg::kEntryThreads), sizeof(double) * vg::lm_entry_solve_lds_doubles(G), st, r2);
                else
                    hipLaunchKernelGGL(vg::vg_lm_reduced_solve_kernel, dim3(1), dim3(G <= 64 ? vg::kWave : vg::kLmThreads), solve_lds, st, r2);
                VG_HIP(hipGetLastError());
            }
            vg::BacksubArgs ba;
            ba.s = sa;
            ba.dg = d_dg.p;
            ba.pose_param = d_pose_param.p;
            ba.gcol_param = d_gcol_param.p;
            ba.delta = d_delta.p;
            ba.scal = d_scal.p;
            ba.gmax_bits = d_gmax.p;
            ba.x = xbuf[par];
            ba.xg = d_xg.p;
            ba.lo = d_glo.p;
            ba.hi = d_ghi.p;
            ba.x_new = xbuf[1 - par];   // the step is applied where it is computed: no separate launch
            ba.fold = fold_frames ? d_fold.p : nullptr;   // ... and so are the candidate's frames
            ba.fold_gcol = d_fold_gcol.p;
            if (n_poses || G) {
                const unsigned int bs_grid = n_bs_groups ? n_bs_groups : 1u;
                if (fold_solve) {
                    r2.S = nullptr;  // the damped matrix in every workgroup's own LDS
                    r2.one_wave = vgi::debug_hook(vgi::kHookSolverOneWaveFold) ? 1 : 0;
                    // kJ = columns per lane of a pose's 16-lane group: 1 up to 15 global columns (every mono problem), 2 up to 31
                    const size_t fold_lds = sizeof(double) * std::max(vg::lm_entry_solve_lds_doubles(G), 2 * (size_t)G * G + 4 * (size_t)G + 2);
                    const bool fr = ba.fold != nullptr;   // the instantiation that also builds the candidate's frames
                    if (G < 16) {
                        if (fr) hipLaunchKernelGGL((vg::vg_backsub_solve_kernel<1, true>), dim3(bs_grid), dim3(vg::kBsThreads), fold_lds, st, ba, r2);
                        else hipLaunchKernelGGL((vg::vg_backsub_solve_kernel<1, false>), dim3(bs_grid), dim3(vg::kBsThreads), fold_lds, st, ba, r2);
                    } else {
                        if (fr) hipLaunchKernelGGL((vg::vg_backsub_solve_kernel<2, true>), dim3(bs_grid), dim3(vg::kBsThreads), fold_lds, st, ba, r2);
                        else hipLaunchKernelGGL((vg::vg_backsub_solve_kernel<2, false>), dim3(bs_grid), dim3(vg::kBsThreads), fold_lds, st, ba, r2);
                    }
                } else vg::launch_backsub(st, G, bs_grid, ba);
                VG_HIP(hipGetLastError());
            }
            p->gram_gate = gate;
            p->gram_gate_expect = par;
            // several ranks: the step's scalar sums are part of the evaluation's packed all-reduce (one rank: the accept kernel sums them)
            const int re = enqueue_evaluate(xbuf[1 - par], gset[1 - par], fold_frames && n_poses > 0, n_bs_groups && multi_rank);
            p->gram_gate = nullptr;
            if (re != VG_OK) return re;
            vg::LmAcceptArgs a2 = aa;
            a2.gate_expect = gated ? par : -1;
            slot = next_slot();
            arm_slot(slot, a2);
            hipLaunchKernelGGL(vg::vg_lm_accept_kernel, dim3(1), dim3(vg::kLmThreads), accept_lds, st, a2);
            VG_HIP(hipGetLastError());
            return queue_state(slot);
        };
        auto wait_state = [&](int slot) -> int {
            if (!spin_wait) {
                VG_HIP(hipEventSynchronize(slots.ev[slot]));
                return VG_OK;
            }
            const double t_spin = now_s();
            unsigned long spins = 0;
            while (slots.seq[slot] != slots.expect[slot]) {
                if ((++spins & 0xfffff) == 0 && now_s() - t_spin > 30.) {   // the device is gone or the launch failed: do not hang
                    VG_HIP(hipStreamSynchronize(st));
                    if (slots.seq[slot] != slots.expect[slot]) return fail(VG_ERR_STATE, "the accept kernel of an LM iteration never reported");
                }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            return VG_OK;
        };
        if (t_arena) VG_TRY(t_arena->flush(st));  // every table of the set-up in one asynchronous copy
        mark("device-loop state");
        const double t_loop = now_s();  // everything before: allocation and upload of the problem's solver state
        VG_TRY(launch_init());   // clears, starting point into both parameter buffers, initial state
        VG_TRY(enqueue_evaluate(xbuf[0], gset[0]));
        int parity = 0, pending = next_slot(), iter = 0;
        arm_slot(pending, aa);
        hipLaunchKernelGGL(vg::vg_lm_accept_kernel, dim3(1), dim3(vg::kLmThreads), accept_lds, st, aa);
        VG_HIP(hipGetLastError());
        aa.init = 0;
        if (opt.max_num_iterations >= 1) VG_TRY(queue_iteration(parity, speculate, pending));
        else VG_TRY(queue_state(pending));
        const vg::LmState *Sp = slots.p + pending;
        bool printed_header = false;
        // Near the end no iteration is queued ahead: the iteration queued behind the LAST one still runs its six launches as
        // closed-gate kernels (27 us at 10 k images, in front of the copy of the result: 5 % of the solve).  LM converges
        // quadratically at the tail, so once the last known step changed the cost by less than 1e-9 of it the iteration in flight
        // is the last or the one before it; not speculating past it costs one launch latency (~8 us) if it was not.
        // (vg_debug_set("solver_no_speculation", 2): always speculate, for A/B.)
        const bool always_speculate = vgi::debug_hook(vgi::kHookSolverNoSpeculation) == 2;
        double last_rel_change = 1.;
        for (iter = 1; iter <= opt.max_num_iterations; iter++) {
            int spec = -1;
            const bool near_end = !always_speculate && last_rel_change <= 1e-9;
            if (speculate && !near_end && iter < opt.max_num_iterations) VG_TRY(queue_iteration(parity ^ 1, true, spec));
            VG_TRY(wait_state(pending));  // the one wait of the iteration; the GPU already holds the next one
            Sp = slots.p + pending;
            const vg::LmState &S = *Sp;
            last_rel_change = (S.step_ok && S.cost2 > 0.) ? std::fabs(2. * S.cost_change) / S.cost2 : 1.;
            if (opt.verbose) {
                if (!printed_header)
                    std::printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n%4d  %.6e\n", 0, 0.5 * S.cost2_init);
                printed_header = true;
                std::printf("%4d  %.6e  %10.3e  %10.3e  %9.3e  %9.3e  %9.3e %s\n", iter, 0.5 * S.cost2, S.cost_change, S.grad_max,
                            S.step_norm, S.rho, S.radius, S.accepted ? "" : (S.done && S.term <= VG_TERM_CONVERGENCE_PARAMETER ? "(converged)" : "(rejected)"));
            }
            if (S.accepted) parity ^= 1;
            if (S.done || iter == opt.max_num_iterations) {
                if (!S.done) iter++;  // ran out of iterations
                break;
            }
            if (S.accepted && spec >= 0) pending = spec;                       // the queued iteration is the real one
            else VG_TRY(queue_iteration(parity, speculate, pending));         // rejected: what was queued has skipped itself
        }
        VG_TRY(wait_state(pending));
        d_x.p = xbuf[parity];       // DevBuf handles: keep ownership of both buffers, current one in d_x
        d_xc.p = xbuf[1 - parity];
        const double initial_cost = 0.5 * Sp->cost2_init;
        final_state = *Sp;
        const vg::LmState &S = final_state;
        char msg[160] = "";
        int term = S.done ? S.term : VG_TERM_NO_CONVERGENCE;
        if (iter > opt.max_num_iterations) {
            iter = opt.max_num_iterations;
            std::snprintf(msg, sizeof msg, "maximum number of iterations reached");
        } else if (term == VG_TERM_CONVERGENCE_GRADIENT)
            std::snprintf(msg, sizeof msg, "gradient tolerance reached: max norm %.3e <= %.3e", S.grad_max, opt.gradient_tolerance);
        else if (term == VG_TERM_CONVERGENCE_PARAMETER) std::snprintf(msg, sizeof msg, "parameter tolerance reached: |step| %.3e", S.step_norm);
        else if (term == VG_TERM_CONVERGENCE_FUNCTION)
            std::snprintf(msg, sizeof msg, "function tolerance reached: |cost change| / cost = %.3e",
                          S.cost2 > 0 ? std::fabs(2. * S.cost_change) / S.cost2 : 0.);   // (the solve ends at the current point: cost2 is its cost)
        else if (term == VG_TERM_RADIUS_TOO_SMALL) std::snprintf(msg, sizeof msg, "trust region radius below %.1e", opt.min_trust_region_radius);
        else if (term == VG_TERM_FAILURE) {
            iter = 0;
            std::snprintf(msg, sizeof msg, "the cost at the starting point is not finite (NaN / Inf in the residuals)");
        }
        if (S.n_bad) {
            const size_t len = std::strlen(msg);
            std::snprintf(msg + len, sizeof msg - len, "%s%d pose block(s) not positive definite", len ? "; " : "", S.n_bad);
        }
        VG_HIP(hipMemcpyAsync(p->d_params, d_x.p, sizeof(double) * (size_t)n_params, hipMemcpyDeviceToDevice, st));
        VG_HIP(hipStreamSynchronize(st));
        p->frames_stale = true;
        if (sum) {
            std::memset(sum, 0, sizeof *sum);
            sum->initial_cost = initial_cost;
            sum->final_cost = 0.5 * S.cost2;
            sum->num_iterations = iter;
            sum->num_successful_steps = S.n_success;
            sum->termination = term;
            sum->gradient_max_norm = S.grad_max;
            sum->final_radius = S.radius;
            sum->total_seconds = now_s() - t_start;
            sum->host_seconds = t_loop - t_start;            // set-up: buffers, index tables, uploads
            sum->evaluate_seconds = now_s() - t_loop;        // the iterations (device resident)
            sum->num_global_columns = G;
            sum->num_pose_blocks = n_poses;
            std::snprintf(sum->message, sizeof sum->message, "%s", msg);
        }
        return VG_OK;
    }

    if (t_arena) VG_TRY(t_arena->flush(st));
    VG_TRY(launch_init());
    // values of the global columns at the starting point
    // (ONE copy of the span they lie in -- the global blocks are neighbours in the parameter vector -- not a blocking copy per
    // column: 45 x 20 us in front of the rig's first iteration, rocprofv3 trace)
    if (G) {
        long long lo_p = gcol_param[0], hi_p = gcol_param[0];
        for (int a2 = 1; a2 < G; a2++) {
            lo_p = gcol_param[a2] < lo_p ? gcol_param[a2] : lo_p;
            hi_p = gcol_param[a2] > hi_p ? gcol_param[a2] : hi_p;
        }
        std::vector<double> span((size_t)(hi_p - lo_p + 1));
        VG_HIP(hipMemcpyAsync(span.data(), p->d_params + lo_p, sizeof(double) * span.size(), hipMemcpyDeviceToHost, st));
        VG_HIP(hipStreamSynchronize(st));
        for (int a2 = 0; a2 < G; a2++) h_xg[a2] = span[(size_t)(gcol_param[a2] - lo_p)];
    }
    std::vector<double> h_xcur(h_xg);  // global values at the CURRENT point (h_xg is refreshed only after the reduced solve)

    DevBuf<double> *cur = gramA, *cand = gramB;
    vg::SolveDatasetDev *ds_cur = d_dsA.p, *ds_cand = d_dsB.p;
    double cost2 = 0., cost2_c = 0.;
    VG_TRY(evaluate(d_x.p, cur, U, gg, cost2));
    {
        std::vector<double> pack(U);
        pack.insert(pack.end(), gg.begin(), gg.end());
        pack.push_back(cost2);
        VG_TRY(allreduce(pack));
        std::copy(pack.begin(), pack.begin() + (size_t)G * G, U.begin());
        std::copy(pack.begin() + (size_t)G * G, pack.begin() + (size_t)G * G + G, gg.begin());
        cost2 = pack.back();
        add_priors(h_xg, U, gg, cost2);
    }
    for (auto &c2 : coupled) {
        VG_HIP(hipMemcpy(c2.x.data(), d_x.p + c2.param_off, sizeof(double) * c2.x.size(), hipMemcpyDeviceToHost));
        cost2 += c2.cost2(c2.x, h_xg.data());
        c2.add_global_terms(c2.x, h_xg.data(), G, U, gg);
    }
    double radius = opt.initial_trust_region_radius, decrease_factor = 2.;
    int iter = 0, n_success = 0, term = VG_TERM_NO_CONVERGENCE;
    double grad_max = 0.;
    const double initial_cost = 0.5 * cost2;
    char msg[160] = "";
    if (opt.verbose) std::printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n%4d  %.6e\n", 0, initial_cost);

    if (!std::isfinite(cost2)) {  // as Ceres: a failed evaluation of the starting point fails the solve
        term = VG_TERM_FAILURE;
        std::snprintf(msg, sizeof msg, "the cost at the starting point is not finite (NaN / Inf in the residuals)");
    }
    std::vector<unsigned char> held;
    std::vector<double> Sw, rw, chol_ws;
    for (iter = 1; term != VG_TERM_FAILURE && iter <= opt.max_num_iterations; iter++) {
        const double mu = 1. / radius;
        // ---- eliminate the poses: rows -> Gram -> S_sub, c
        double t0 = now_s();
        vg::SchurArgs sa;
        sa.ds = ds_cur;
        sa.inv = d_inv.p;
        sa.ref_ptr = d_ref_ptr.p;
        sa.ref_ds = d_ref_ds.p;
        sa.ref_blk = d_ref_blk.p;
        sa.pose_frozen = d_pf.p;
        sa.n_ds = n_ds;
        sa.G = G;
        sa.n_poses = (int)n_poses;
        sa.mu = mu;
        sa.mu_dev = nullptr;
        sa.gate = nullptr;
        sa.gate_expect = 0;
        sa.dmin = opt.min_lm_diagonal;
        sa.dmax = opt.max_lm_diagonal;
        sa.rec = d_rec.p;
        sa.rows = d_rows.p;
        sa.bad = d_bad;
        // the Schur complement is read where the device wrote it when this rank's kernels deliver it straight to pinned memory;
        // otherwise (all-reduce callback, no poses) from the staging vector
        const bool rgram_in_place = host_direct && n_poses > 0;
        if (!rgram_in_place) std::fill(h_rgram.begin(), h_rgram.end(), 0.);
        const double *rg = rgram_in_place ? pin_rgram.p : h_rgram.data();
        const bool schur_spin = host_spin && n_poses > 0 && coupled.empty();
        bool coupled_ok = true;
        if (n_poses) {
            if (coupled.empty()) {
                sa.zero_u64 = d_gmax.p;  // the step's max |g_pose|, cleared here instead of by a memset in front of the back-substitution
                hipLaunchKernelGGL(vg::vg_schur_rows_gram_kernel, dim3(sg_wgs), dim3(vg::kSchurThreads * sg_batches), sg_lds, st, sa, sg_ppw, sg_batches, d_rgroups.p, sg_shared);
            } else {
                VG_HIP(hipMemsetAsync(d_bad, 0, sizeof(double), st));
                hipLaunchKernelGGL(vg::vg_schur_rows_kernel, dim3((unsigned)((n_poses * C + 255) / 256)), dim3(256), 0, st, sa);
            }
            VG_HIP(hipGetLastError());
            // sequences coupled by odometry: raw V / g / W^T come back, the host eliminates the block-tridiagonal
            // system and puts its rows where the per-pose rows would be
            for (auto &c2 : coupled) {
                std::vector<double> hrec((size_t)c2.n * vg::kPoseRec), hraw((size_t)c2.n * 6 * C);
                if (coupled_multi) {  // raw normal-equation pieces of the replicated sequence, summed over the ranks' images
                    VG_TRY(vgc::allreduce_sum(comm, d_rec.p + (size_t)c2.pb * vg::kPoseRec, hrec.size(), st));
                    VG_TRY(vgc::allreduce_sum(comm, d_rows.p + (size_t)c2.pb * 6 * C, hraw.size(), st));
                }
                VG_HIP(hipMemcpyAsync(hrec.data(), d_rec.p + (size_t)c2.pb * vg::kPoseRec, sizeof(double) * hrec.size(),
                                      hipMemcpyDeviceToHost, st));
                VG_HIP(hipMemcpyAsync(hraw.data(), d_rows.p + (size_t)c2.pb * 6 * C, sizeof(double) * hraw.size(),
                                      hipMemcpyDeviceToHost, st));
                VG_HIP(hipStreamSynchronize(st));
                if (!c2.eliminate(hrec.data(), hraw.data(), G, mu, opt.min_lm_diagonal, opt.max_lm_diagonal, h_xcur.data())) {
                    coupled_ok = false;
                    std::fill(c2.Y.begin(), c2.Y.end(), 0.);
                    c2.Y.resize((size_t)c2.n * 6 * C, 0.);
                }
                // the rows enter the Schur complement ONCE: every rank has the same ones, rank 0 contributes them
                if (coupled_multi && comm->rank != 0) {
                    VG_HIP(hipMemsetAsync(d_rows.p + (size_t)c2.pb * 6 * C, 0, sizeof(double) * c2.Y.size(), st));
                } else {
                    VG_HIP(hipMemcpyAsync(d_rows.p + (size_t)c2.pb * 6 * C, c2.Y.data(), sizeof(double) * c2.Y.size(),
                                          hipMemcpyHostToDevice, st));
                }
                VG_HIP(hipStreamSynchronize(st));  // c2.Y may be rewritten before an async copy from pageable memory ends
            }
            if (coupled.empty()) {
                vg::launch_strided_sum(st, d_rgroups.p, sg_wgs, C * C + 1, host_direct ? pin_rgram.p : d_rgram.p,
                                       schur_spin ? host_signal(1) : vg::HostSignal());
            } else {
                VG_TRY(launch_dense_gram(st, d_rows.p, n_rows, C, rows_per_group, n_groups, d_rgroups.p));
                vg::launch_strided_sum(st, d_rgroups.p, n_groups, C * C, d_rgram.p);
            }
            VG_HIP(hipGetLastError());
        } else if (comm && comm->n_ranks > 1) {
            VG_HIP(hipMemsetAsync(d_rgram.p, 0, sizeof(double) * h_rgram.size(), st));  // a rank without poses still joins the sum
        }
        if (n_poses || (comm && comm->n_ranks > 1)) {
            if (!host_direct) {
                VG_TRY(vgc::allreduce_sum(comm, d_rgram.p, h_rgram.size(), st));  // Schur complement of the poses of all ranks
                VG_HIP(hipMemcpyAsync(pin_rgram.p, d_rgram.p, sizeof(double) * h_rgram.size(), hipMemcpyDeviceToHost, st));
            }
            if (schur_spin) VG_TRY(host_wait(1));
            else VG_HIP(hipStreamSynchronize(st));
            if (!rgram_in_place) std::memcpy(h_rgram.data(), pin_rgram.p, sizeof(double) * h_rgram.size());
        }
        if (opt.allreduce) VG_TRY(allreduce(h_rgram));   // (host_direct excludes the callback: rg stays valid)
        // poses whose damped 6 x 6 block was not positive definite (NaN / Inf in their Gram block): the step is invalid
        // as a whole -- rejected like a failed factorisation of the reduced system, and counted.  The count is the one
        // summed over ALL ranks (last slot of the buffer): a rank-local decision here would make this rank skip the
        // collectives of the candidate evaluation while the others enter them.
        if (rg[(size_t)C * C] > 0.) {
            coupled_ok = false;
            n_bad_pose_blocks += (long long)rg[(size_t)C * C];
        }
        t_schur += now_s() - t0;

        // ---- reduced system on the host
        t0 = now_s();
        // (rows 0 .. G - 1 of the Schur complement's lower triangle and its last ROW, which is its last column: half the cache
        //  lines of what the device wrote)
        for (int a2 = 0; a2 < G; a2++) {
            for (int b2 = 0; b2 <= a2; b2++) {
                const double r2 = rg[(size_t)a2 * C + b2];
                S[(size_t)a2 * G + b2] = U[(size_t)a2 * G + b2] - r2;
                if (a2 != b2) S[(size_t)b2 * G + a2] = U[(size_t)b2 * G + a2] - r2;
            }
            const double dd = U[(size_t)a2 * G + a2];
            S[(size_t)a2 * G + a2] += mu * (dd < opt.min_lm_diagonal ? opt.min_lm_diagonal : (dd > opt.max_lm_diagonal ? opt.max_lm_diagonal : dd));
            rhs[a2] = -gg[a2] + rg[(size_t)G * C + a2];
        }
        // Constant blocks, and the active set of the box bounds: a parameter sitting ON a bound whose step points
        // outwards is held for this iteration (its row / column leave the reduced system -- the Schur complement of
        // the constrained problem is exactly that sub-matrix).  Without this the projected step keeps "spending" its
        // decrease on a coordinate that cannot move, the gain ratio collapses and the radius shrinks to nothing.
        held.assign(gfrozen.begin(), gfrozen.end());   // (held, Sw, rw, chol_ws: allocated once, in front of the loop)
        bool step_ok = coupled_ok;
        for (int pass = 0; step_ok && pass <= G; pass++) {
            Sw = S;
            rw = rhs;
            for (int a2 = 0; a2 < G; a2++)
                if (held[a2]) {
                    for (int b2 = 0; b2 < G; b2++) Sw[(size_t)a2 * G + b2] = Sw[(size_t)b2 * G + a2] = 0.;
                    Sw[(size_t)a2 * G + a2] = 1.;
                    rw[a2] = 0.;
                }
            step_ok = G == 0 || chol_solve(G, Sw.data(), rw.data(), dg.data(), chol_ws);
            bool changed = false;
            if (step_ok && opt.use_bounds)
                for (int a2 = 0; a2 < G; a2++) {
                    if (held[a2]) continue;
                    const double l2 = glo[(size_t)a2], h2 = ghi[(size_t)a2];
                    if ((h_xcur[a2] <= l2 && dg[a2] < 0.) || (h_xcur[a2] >= h2 && dg[a2] > 0.)) {
                        held[a2] = 1;
                        changed = true;
                    }
                }
            if (!changed) break;
        }
        t_host += now_s() - t0;

        double model_change = 0., step2 = 0., cost_change = 0., rho = 0.;
        if (step_ok) {
            // ---- back-substitute, apply, evaluate the candidate
            t0 = now_s();
            if (G) {
                std::memcpy(pin_small.p, dg.data(), sizeof(double) * G);
                if (!host_direct) VG_HIP(hipMemcpyAsync(d_dg.p, pin_small.p, sizeof(double) * G, hipMemcpyHostToDevice, st));
            }
            if (!(n_poses && coupled.empty())) VG_HIP(hipMemsetAsync(d_gmax.p, 0, sizeof(unsigned long long), st));  // else: cleared by the rows kernel
            vg::BacksubArgs ba;
            ba.s = sa;
            ba.dg = host_direct ? pin_small.p : d_dg.p;   // the reduced step: read where the host wrote it
            ba.pose_param = d_pose_param.p;
            ba.gcol_param = d_gcol_param.p;
            ba.delta = d_delta.p;
            ba.scal = d_scal.p;
            ba.gmax_bits = d_gmax.p;
            ba.x = d_x.p;
            double *ps = pin_small.p + G;  // [gmax 1 | xg G]
            ba.xg = host_direct ? ps + 1 : d_xg.p;   // current values of the global columns, for the host
            ba.lo = d_glo.p;
            ba.hi = d_ghi.p;
            ba.x_new = d_xc.p;   // host-eliminated sequences overwrite their poses below
            ba.fold = fold_frames ? d_fold.p : nullptr;   // the candidate's frames come out of the same launch
            ba.fold_gcol = d_fold_gcol.p;
            if (n_poses || G) {  // G <= kBsThreads: one workgroup is enough for the global columns alone
                const unsigned int bs_grid = n_bs_groups ? n_bs_groups : 1u;
                vg::launch_backsub(st, G, bs_grid, ba);
                VG_HIP(hipGetLastError());
            }
            // (the fixed-order sum of the back-substitution's per-workgroup partials and, host_direct, max |g_pose| to the host:
            //  with the candidate's evaluation below)
            double host_scal[5] = {0., 0., 0., 0., 0.};
            for (auto &c2 : coupled) {
                std::vector<double> dp;
                double sc[5];
                c2.backsub(dg.data(), G, dp, sc);
                for (int k = 0; k < 4; k++) host_scal[k] += sc[k];
                host_scal[4] = sc[4] > host_scal[4] ? sc[4] : host_scal[4];
                VG_HIP(hipMemcpyAsync(d_delta.p + c2.param_off, dp.data(), sizeof(double) * dp.size(), hipMemcpyHostToDevice, st));
                VG_HIP(hipStreamSynchronize(st));
            }
            // the back-substitution kernel wrote the candidate of every global column and of every pose it owns; the
            // poses of host-eliminated sequences (unbounded) take their steps here
            for (auto &c2 : coupled) {
                const long long n2 = (long long)c2.n * 6;
                hipLaunchKernelGGL(vg::vg_apply_step_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st,
                                   (const double *)d_x.p + c2.param_off, (const double *)d_delta.p + c2.param_off, n2,
                                   d_xc.p + c2.param_off);
                VG_HIP(hipGetLastError());
            }
            // without poses the five scalar sums (tail of the sums block) stay at the zeros they were initialised with, and so
            // does max |g_pose|
            if (!host_direct) VG_HIP(hipMemcpyAsync(ps, d_small.p, sizeof(double) * (1 + (size_t)G), hipMemcpyDeviceToHost, st));
            t_schur += now_s() - t0;
            // No wait here: the candidate evaluation does not depend on these scalars, it is queued right behind the
            // step on the same stream, and its own read-back synchronises once for both (one host round trip per
            // iteration less; the wait is booked under "evaluate").
            VG_TRY(evaluate(d_xc.p, cand, Uc, ggc, cost2_c, fold_frames && n_poses > 0, n_bs_groups > 0,
                            host_direct ? reinterpret_cast<unsigned long long *>(ps) : nullptr));

            // |x|^2 of this rank's pose parameters (summed over ranks below) and of the replicated global block
            for (int a2 = 0; a2 < G; a2++) h_xg[a2] = ps[1 + a2];
            const double *sc = pin_sums.p + n_sums;  // scalar sums of the step, already summed over ranks with an RCCL communicator
            double xg2 = 0.;
            for (int a2 = 0; a2 < G; a2++) xg2 += h_xg[a2] * h_xg[a2];
            double gdp = sc[0] + host_scal[0], ddp = sc[1] + host_scal[1], dp2 = sc[2] + host_scal[2],
                   gp2 = sc[3] + host_scal[3], xp2 = sc[4], gmax_p = ps[0] > host_scal[4] ? ps[0] : host_scal[4];
            // global values of the candidate: clamp(x + dg), as vg_apply_step_kernel does
            std::vector<double> xg_c(G);
            for (int a2 = 0; a2 < G; a2++) {
                const double v = h_xg[a2] + dg[a2], l2 = glo[(size_t)a2], h2 = ghi[(size_t)a2];
                xg_c[a2] = v < l2 ? l2 : (v > h2 ? h2 : v);
            }
            for (auto &c2 : coupled) {
                VG_HIP(hipMemcpy(c2.xc.data(), d_xc.p + c2.param_off, sizeof(double) * c2.xc.size(), hipMemcpyDeviceToHost));
                cost2_c += c2.cost2(c2.xc, xg_c.data());
                if (coupled_multi) {  // the replicated poses entered the summed |x|^2 once per rank
                    double x2 = 0.;
                    for (double v : c2.x) x2 += v * v;
                    xp2 -= (double)(comm->n_ranks - 1) * x2;
                }
            }
            {
                if (opt.allreduce) {   // (no callback: nothing to pack, sum and unpack)
                std::vector<double> pack(Uc);
                pack.insert(pack.end(), ggc.begin(), ggc.end());
                pack.push_back(cost2_c);
                pack.push_back(gdp);
                pack.push_back(ddp);
                pack.push_back(dp2);
                pack.push_back(gp2);
                pack.push_back(xp2);
                VG_TRY(allreduce(pack));
                size_t o = (size_t)G * G;
                std::copy(pack.begin(), pack.begin() + o, Uc.begin());
                std::copy(pack.begin() + o, pack.begin() + o + G, ggc.begin());
                o += G;
                cost2_c = pack[o];
                gdp = pack[o + 1];
                ddp = pack[o + 2];
                dp2 = pack[o + 3];
                gp2 = pack[o + 4];
                xp2 = pack[o + 5];
                }
                // The callback only sums.  With several ranks every rank must take the same branches, so the
                // pose part of the gradient max-norm is replaced by its (summable) 2-norm, an upper bound:
                // the gradient test can only fire later than Ceres' max-norm test, never earlier.
                if (multi_rank) gmax_p = std::sqrt(gp2);
                if (!p->priors.empty()) add_priors(xg_c, Uc, ggc, cost2_c);
                for (auto &c2 : coupled) c2.add_global_terms(c2.xc, xg_c.data(), G, Uc, ggc);
            }
            double gdg = 0., ddg = 0., dg2 = 0., gmax_g = 0.;
            for (int a2 = 0; a2 < G; a2++) {
                if (gfrozen[a2]) continue;
                const double dd = U[(size_t)a2 * G + a2];
                const double dcl = dd < opt.min_lm_diagonal ? opt.min_lm_diagonal : (dd > opt.max_lm_diagonal ? opt.max_lm_diagonal : dd);
                gdg += gg[a2] * dg[a2];
                ddg += dcl * dg[a2] * dg[a2];
                dg2 += dg[a2] * dg[a2];
                // projected gradient for bounded parameters: |Project(x - g) - x|
                const double xv = h_xg[a2];
                double xg = xv - gg[a2];
                const double l2 = glo[(size_t)a2], h2 = ghi[(size_t)a2];
                xg = xg < l2 ? l2 : (xg > h2 ? h2 : xg);
                gmax_g = std::fabs(xg - xv) > gmax_g ? std::fabs(xg - xv) : gmax_g;
            }
            grad_max = gmax_g > gmax_p ? gmax_g : gmax_p;
            // model decrease of the exact LM step: 1/2 delta^T (mu D delta - g)
            model_change = 0.5 * (mu * (ddg + ddp) - (gdg + gdp));
            step2 = dg2 + dp2;
            cost_change = 0.5 * (cost2 - cost2_c);
            rho = model_change > 0. ? cost_change / model_change : -1.;

            if (grad_max <= opt.gradient_tolerance) {
                term = VG_TERM_CONVERGENCE_GRADIENT;
                std::snprintf(msg, sizeof msg, "gradient tolerance reached: max norm %.3e <= %.3e", grad_max, opt.gradient_tolerance);
                break;
            }
            const double xn2 = xg2 + xp2;  // identical on every rank
            if (std::sqrt(step2) <= opt.parameter_tolerance * (std::sqrt(xn2) + opt.parameter_tolerance)) {
                term = VG_TERM_CONVERGENCE_PARAMETER;
                std::snprintf(msg, sizeof msg, "parameter tolerance reached: |step| %.3e", std::sqrt(step2));
                break;
            }
            // Function tolerance: Ceres tests |cost change| of EVERY evaluated candidate of a valid step, before it decides whether
            // the step is accepted (trust_region_minimizer.cc: the "function tolerance reached" block / FunctionToleranceReached()
            // sits in front of the relative-decrease test), and returns at the CURRENT point.  With the reference's 1e-15 this is
            // what ends the cascade of rejected noise-level steps at the tail of a solve after three or four radius reductions
            // instead of the eight the parameter tolerance needs; until round 4 the test ran for accepted steps only.
            if (model_change > 0. && std::isfinite(cost2_c) && std::fabs(cost2 - cost2_c) <= opt.function_tolerance * cost2) {
                term = VG_TERM_CONVERGENCE_FUNCTION;
                std::snprintf(msg, sizeof msg, "function tolerance reached: |cost change| / cost = %.3e",
                              cost2 > 0 ? std::fabs(cost2 - cost2_c) / cost2 : 0.);
                break;
            }
        }
        const bool success = step_ok && std::isfinite(cost2_c) && rho > opt.min_relative_decrease;
        if (opt.verbose)
            std::printf("%4d  %.6e  %10.3e  %10.3e  %9.3e  %9.3e  %9.3e %s\n", iter, 0.5 * (success ? cost2_c : cost2), cost_change,
                        grad_max, std::sqrt(step2), rho, radius, success ? "" : "(rejected)");
        if (success) {
            n_success++;
            std::swap(cur, cand);
            std::swap(ds_cur, ds_cand);
            std::swap(d_x.p, d_xc.p);
            for (auto &c2 : coupled) c2.x.swap(c2.xc);
            for (int a2 = 0; a2 < G; a2++) {  // what vg_apply_step_kernel wrote: clamp(x + dg)
                const double v = h_xg[a2] + dg[a2], l2 = glo[(size_t)a2], h2 = ghi[(size_t)a2];
                h_xcur[a2] = v < l2 ? l2 : (v > h2 ? h2 : v);
            }
            U.swap(Uc);
            gg.swap(ggc);
            cost2 = cost2_c;
            const double f = 1. - std::pow(2. * rho - 1., 3);
            radius = radius / (f > 1. / 3. ? f : 1. / 3.);
            radius = radius > opt.max_trust_region_radius ? opt.max_trust_region_radius : radius;
            decrease_factor = 2.;
        } else {
            radius /= decrease_factor;
            decrease_factor *= 2.;
            if (radius < opt.min_trust_region_radius) {
                term = VG_TERM_RADIUS_TOO_SMALL;
                std::snprintf(msg, sizeof msg, "trust region radius below %.1e", opt.min_trust_region_radius);
                break;
            }
        }
    }
    if (term == VG_TERM_FAILURE) iter = 0;
    else if (iter > opt.max_num_iterations) {
        iter = opt.max_num_iterations;
        std::snprintf(msg, sizeof msg, "maximum number of iterations reached");
    }
    if (n_bad_pose_blocks) {
        const size_t len = std::strlen(msg);
        std::snprintf(msg + len, sizeof msg - len, "%s%lld pose block(s) not positive definite", len ? "; " : "", n_bad_pose_blocks);
    }
    VG_HIP(hipMemcpyAsync(p->d_params, d_x.p, sizeof(double) * (size_t)n_params, hipMemcpyDeviceToDevice, st));
    VG_HIP(hipStreamSynchronize(st));
    p->frames_stale = true;  // whatever frames are in HBM belong to some candidate point, not to the solution
    if (sum) {
        std::memset(sum, 0, sizeof *sum);
        sum->initial_cost = initial_cost;
        sum->final_cost = 0.5 * cost2;
        sum->num_iterations = iter;
        sum->num_successful_steps = n_success;
        sum->termination = term;
        sum->gradient_max_norm = grad_max;
        sum->final_radius = radius;
        sum->total_seconds = now_s() - t_start;
        sum->evaluate_seconds = t_eval;
        sum->schur_seconds = t_schur;
        sum->host_seconds = t_host;
        sum->num_global_columns = G;
        sum->num_pose_blocks = n_poses;
        std::snprintf(sum->message, sizeof sum->message, "%s", msg);
    }
#undef VG_TRY
    return VG_OK;
}

}  // extern "C"
