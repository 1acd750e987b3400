// vg_lm_solve.hpp -- struct LmSolve: one Levenberg-Marquardt solve with per-pose Schur elimination (kernels: vg_solver.hpp,
// vg_solver_device.hpp), the role of ceres::Solve at src/calibration/unified_calibration.cpp:53 for problems made of
// GenericProjectionJac blocks, priors and odometry blocks.  vg_problem_solve (vg_solver_impl.hpp) constructs one and calls run().
//
// The object holds what the phases of a solve share -- the column / pose bookkeeping, the index tables, every device and pinned
// buffer (views into ONE device block and ONE pinned block, vg_solver_memory.hpp), the evaluation routine -- and its member
// functions are the phases:
//   set-up        setup_columns, setup_coupled, setup_dataset_maps, setup_device_state, setup_sum_tables
//   evaluation    enqueue_evaluate (Gram blocks of every dataset at a device parameter buffer, their fixed-order sums, the ONE
//                 collective of an evaluation), evaluate (+ read-back and assembly of the global block on the host)
//   device loop   run_device_loop: the trust-region state lives on the device, an iteration is a fixed sequence of six launches
//                 (DeviceLoop::queue_iteration) queued speculatively behind device-side gates; problems of grid blocks only,
//                 up to 32 global columns
//   host loop     run_host_loop: wide reduced systems (the rig), priors, odometry-coupled sequences, the all-reduce callback;
//                 per iteration host_eliminate_poses -> host_reduced_solve -> host_step_and_evaluate -> host_accept
//   multi-rank    the collectives sit inside enqueue_evaluate (summed Gram blocks + step scalars) and behind the pose elimination
//                 (Schur complement + count of bad pose blocks): vgc::allreduce_sum on the problem's stream
// Until round 5 this was one function of ~1 350 lines with the two loops as fragments #included into its body.
#pragma once

#include <memory>

namespace {

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define VG_TRY(e) do { if ((rc = (e)) != VG_OK) return rc; } while (0)

struct LmSolve {
    // ------------------------------------------------------------------------------------------ what is solved, with what
    vg_problem *const p;
    const vg_solve_options opt;
    vg_solve_summary *const sum;
    hipStream_t st = nullptr;
    double t_start = 0., t_eval = 0., t_schur = 0., t_host = 0., t_mark = 0.;
    bool trace_setup = false;  // measurement hook: where the set-up time goes

    // ------------------------------------------------------------------------------------------ columns and poses
    int n_ds = 0;
    int64_t n_params = 0;
    std::vector<int> cam_goff, tf_goff, pb_goff;   // first global column of a camera / global transform / parameter block
    std::vector<int64_t> tf_pbase;                 // first pose block of a sequence transform
    int G = 0, C = 1;                              // global columns; C = G + 1 (the right-hand side rides along)
    int64_t n_poses = 0;
    std::vector<unsigned char> gfrozen, pose_frozen;   // pose_frozen: 0 free, 1 constant, 2 eliminated on the host (coupled)
    std::vector<long long> gcol_param, pose_param;
    std::vector<double> glo, ghi;                  // box bounds of the global columns (intrinsics only)
    std::vector<CoupledSeq> coupled;               // sequences coupled by odometry blocks / unary priors: host elimination
    const vg_comm *comm = nullptr;
    bool multi_rank = false, coupled_multi = false, device_loop = false, host_direct = false, host_spin = false;

    // per dataset: local -> global column map, pose column offset, pose references (CSR over the poses)
    std::vector<std::vector<int>> lmap;
    std::vector<int> inv, Wd, pose_off, seq_tf;
    int Wmax = 1;
    std::vector<int> ref_ptr, ref_ds, ref_blk;

    // ------------------------------------------------------------------------------------------ launch geometry
    unsigned int n_rows = 0, rows_per_group = 96, n_groups = 0, n_slabs = 0, sg_wgs = 0, n_bs_groups = 0;
    int sg_ppw = 1, sg_batches = 1, sg_shared = 1;
    size_t sg_lds = 0, n_sums = 0, n_pack = 0;

    // ------------------------------------------------------------------------------------------ memory (the arena first: it is released last)
    std::unique_ptr<ArenaScope> arena;
    std::vector<DevBuf<double>> gramA_v, gramB_v;   // the two alternating sets of per-image Gram blocks, one buffer per dataset
    DevBuf<double> *gramA = nullptr, *gramB = nullptr;
    DevBuf<double> d_sums, d_x, d_xc, d_delta, d_glo, d_ghi, d_rec, d_rows, d_rgroups, d_rgram, d_dg, d_scal, d_small;
    DevBuf<vg::SolveDatasetDev> d_dsA, d_dsB;
    DevBuf<int> d_inv, d_ref_ptr, d_ref_ds, d_ref_blk;
    DevBuf<unsigned char> d_pf;
    DevBuf<long long> d_pose_param, d_gcol_param;
    bool fold_frames = false;                       // the back-substitution launch also builds the candidate's frames
    DevBuf<vg::PrepDataset> d_fold;
    DevBuf<int> d_fold_gcol;
    double *d_bad = nullptr, *d_scal_sum = nullptr, *d_xg = nullptr, *sums_out = nullptr;
    unsigned long long *d_gmax = nullptr;
    // the current and the candidate parameter vector: raw pointers into d_x / d_xc that swap on an accepted step (the DevBuf
    // handles keep their own allocation whatever the parity is)
    double *x_cur = nullptr, *x_cand = nullptr;
    vg::SolverInitArgs init;                        // what ONE launch clears / copies before the first evaluation
    std::vector<double> h_sums, h_rgram, U, gg, Uc, ggc, S, rhs, dg, h_xg;
    PinnedBuf pin_sums, pin_rgram, pin_small, pin_seq;
    long long n_bad_pose_blocks = 0;
    DevBuf<unsigned int> d_sigcnt;
    unsigned long long seq_issued[2] = {0ull, 0ull};

    // fixed-order sums of the Gram blocks
    std::vector<DevBuf<double>> sum_partials, wg_partials;
    std::vector<double *> wg_partials_ptr;
    DevBuf<vg::SumDataset> d_sumA, d_sumB;
    DevBuf<vg::PartialSumDataset> d_psum;
    unsigned int sum_slab_blocks = 0, sum_final_blocks = 0, psum_blocks = 0;
    int n_psum = 0;
    bool use_partials = false;

    LmSolve(vg_problem *problem, const vg_solve_options &options, vg_solve_summary *summary) : p(problem), opt(options), sum(summary) {}
    ~LmSolve()
    {
        // whatever is still queued may write into the arena's blocks (speculative iterations, the accept kernel's pinned state):
        // nothing is handed back to the process-wide cache before the stream is idle; blocks of a stream that cannot be
        // drained are freed, not cached
        if (arena && st_valid && hipStreamSynchronize(st) != hipSuccess) {
            (void)hipGetLastError();
            arena->discard = true;
        }
    }
    LmSolve(const LmSolve &) = delete;
    LmSolve &operator=(const LmSolve &) = delete;
    bool st_valid = false;

    void mark(const char *what)
    {
        if (!trace_setup) return;
        const double t = now_s();
        std::fprintf(stderr, "[vg_problem_solve] %-28s %8.1f us\n", what, (t - t_mark) * 1e6);
        t_mark = t;
    }

    int run()
    {
        int rc;
        VG_HIP(hipSetDevice(p->device));
        st = p->stream;
        st_valid = true;
        t_start = t_mark = now_s();
        trace_setup = vgi::debug_hook(vgi::kHookSolverTiming) != 0;
        VG_TRY(setup_columns());
        VG_TRY(setup_coupled());
        VG_TRY(setup_dataset_maps());
        mark("host index tables");
        VG_TRY(setup_device_state());
        VG_TRY(setup_sum_tables());
        return device_loop ? run_device_loop() : run_host_loop();
    }

    // ================================================================================================ set-up
    // global columns [cameras | global transforms | parameter blocks], pose blocks, constness, box bounds
    int setup_columns()
    {
        n_ds = (int)p->dss.size();
        n_params = p->n_params;
        cam_goff.assign(p->cams.size(), 0);
        tf_goff.assign(p->tfs.size(), -1);
        tf_pbase.assign(p->tfs.size(), -1);
        G = 0;
        for (size_t c = 0; c < p->cams.size(); c++) { cam_goff[c] = G; G += p->cams[c].K; }
        for (size_t t = 0; t < p->tfs.size(); t++)
            if (p->tfs[t].global) { tf_goff[t] = G; G += 6; }
        pb_goff.assign(p->pblocks.size(), 0);
        for (size_t b = 0; b < p->pblocks.size(); b++) { pb_goff[b] = G; G += p->pblocks[b].size; }
        if (G > 127) return fail(VG_ERR_INVALID_ARGUMENT, "more than 127 global columns are not supported");
        C = G + 1;
        n_poses = 0;
        for (size_t t = 0; t < p->tfs.size(); t++)
            if (!p->tfs[t].global) { tf_pbase[t] = n_poses; n_poses += p->tfs[t].count; }
        if (n_poses > 0x7fffffff / 8) return fail(VG_ERR_INVALID_ARGUMENT, "too many pose blocks");

        gfrozen.assign((size_t)G, 0);
        pose_frozen.assign((size_t)n_poses, 0);
        gcol_param.assign((size_t)G, 0);
        pose_param.assign((size_t)n_poses, 0);
        // box bounds exist for intrinsics only (eucm.h:228-246, ucm.h:199-215, mei.h:287-313, set at unified_calibration.cpp:
        // 621-627), i.e. for global columns: one pair per column, nothing per pose parameter
        glo.assign((size_t)G, -std::numeric_limits<double>::infinity());
        ghi.assign((size_t)G, std::numeric_limits<double>::infinity());
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
        return VG_OK;
    }

    // sequences coupled by OdometryPrior / OdometryCost blocks -> host elimination (pose mode 2); constant elements ("anchor");
    // which communicator; which loop drives the iterations
    int setup_coupled()
    {
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
        comm = opt.comm;
        multi_rank = opt.allreduce != nullptr || (comm && comm->n_ranks > 1);
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
        coupled_multi = !coupled.empty() && multi_rank;
        // which loop drives the iterations; measurement / A-B hooks (vg_debug_set) force a side
        const bool force_host_loop = vgi::debug_hook(vgi::kHookSolverHostLoop) != 0;
        const bool force_device_loop = vgi::debug_hook(vgi::kHookSolverDeviceLoop) != 0;
        device_loop = coupled.empty() && p->priors.empty() && !opt.allreduce && !force_host_loop && n_ds <= vg::kLmMaxDatasets && (G <= 32 || force_device_loop);
        // Host-driven loop on one rank without host-eliminated sequences: what the host reads every iteration (the Schur Gram,
        // the summed Gram blocks, the step's scalars) is WRITTEN INTO PINNED HOST MEMORY by the kernels that produce it, and
        // the reduced step is read from pinned memory by the back-substitution -- no copy or memset command between two kernels
        // (each one is an engine hand-over of ~10 us on this stack; the rig's iteration has six of them otherwise).
        host_direct = !device_loop && !comm && !opt.allreduce && coupled.empty();
        return VG_OK;
    }

    // per dataset: local -> global column map, pose column offset, pose references
    int setup_dataset_maps()
    {
        lmap.assign((size_t)n_ds, std::vector<int>());
        inv.assign((size_t)n_ds * (G ? G : 1), -1);
        Wd.assign((size_t)n_ds, 0);
        pose_off.assign((size_t)n_ds, -1);
        seq_tf.assign((size_t)n_ds, -1);
        Wmax = 1;
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
        ref_ptr.assign((size_t)n_poses + 1, 0);
        for (int d = 0; d < n_ds; d++)
            if (seq_tf[d] >= 0)
                for (int64_t b = 0; b < p->dss[d].n_blocks; b++) ref_ptr[(size_t)(tf_pbase[seq_tf[d]] + p->dss[d].h_seq[(size_t)b]) + 1]++;
        for (int64_t i = 0; i < n_poses; i++) ref_ptr[(size_t)i + 1] += ref_ptr[(size_t)i];
        ref_ds.resize(ref_ptr.back());
        ref_blk.resize(ref_ptr.back());
        std::vector<int> cur(ref_ptr.begin(), ref_ptr.end() - 1);
        for (int d = 0; d < n_ds; d++)
            if (seq_tf[d] >= 0)
                for (int64_t b = 0; b < p->dss[d].n_blocks; b++) {
                    const size_t i = (size_t)(tf_pbase[seq_tf[d]] + p->dss[d].h_seq[(size_t)b]);
                    ref_ds[cur[i]] = d;
                    ref_blk[cur[i]++] = (int)b;
                }
        return VG_OK;
    }

    // launch geometry of the pose elimination, the solve's one device block and one pinned block, every buffer and upload
    int setup_device_state()
    {
        int rc;
        n_rows = (unsigned int)(6 * n_poses);
        rows_per_group = 96;  // 16 poses per wave: enough waves to fill the chip at 5 k poses
        n_groups = n_rows ? (n_rows + rows_per_group - 1) / rows_per_group : 0;
        n_slabs = (n_groups + vg::kSlab - 1) / vg::kSlab;
        // the fused rows + Gram launch (vg_schur_rows_gram_kernel): whole poses per workgroup, `sg_batches` batches of
        // `sg_ppw` poses each so that the partials stay in the hundreds and the rows of a workgroup fit 48 KB of LDS
        sg_ppw = vg::kSchurThreads / (G + 1);
        sg_batches = (int)((n_poses + (int64_t)sg_ppw * 512 - 1) / ((int64_t)sg_ppw * 512));
        sg_batches = sg_batches < 1 ? 1 : (sg_batches > vg::kSchurMaxBatches ? vg::kSchurMaxBatches : sg_batches);
        while (sg_batches > 1 && sizeof(double) * (size_t)sg_batches * sg_ppw * (6 * (G + 2) + 28) + 24 * (size_t)vg::kSchurMaxRefs > 64 * 1024) sg_batches--;   // two 1024-thread workgroups fill a CU: 64 KB each is free
        // + V_i | g_i of every pose of the workgroup, gathered once and shared by the pose's lanes (28 doubles per pose)
        sg_lds = sizeof(double) * ((size_t)sg_batches * sg_ppw * 6 * (G + 2) + (size_t)sg_batches * sg_ppw * 28) + 24 * (size_t)vg::kSchurMaxRefs;
        sg_shared = vgi::debug_hook(vgi::kHookSchurPrivateGather) ? 0 : 1;   // A/B hook: every lane gathers for itself
        if (sg_lds > 48 * 1024)
            VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vg::vg_schur_rows_gram_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sg_lds));
        sg_wgs = (unsigned int)((n_poses + (int64_t)sg_ppw * sg_batches - 1) / ((int64_t)sg_ppw * sg_batches));
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
        arena.reset(new ArenaScope(p->device, dev_need, pin_need, up_need));
        gramA_v = std::vector<DevBuf<double>>((size_t)(n_ds ? n_ds : 1));  // sized once, never resized
        gramB_v = std::vector<DevBuf<double>>((size_t)(n_ds ? n_ds : 1));
        gramA = gramA_v.data();
        gramB = gramB_v.data();
        std::vector<vg::SolveDatasetDev> hdsA(n_ds), hdsB(n_ds);
        for (int d = 0; d < n_ds; d++) {
            const size_t n = (size_t)p->dss[d].n_blocks * Wd[d] * Wd[d];
            if ((rc = gramA[d].alloc(n)) != VG_OK || (rc = gramB[d].alloc(n)) != VG_OK) return rc;
            hdsA[d] = {gramA[d].p, Wd[d], pose_off[d]};
            hdsB[d] = {gramB[d].p, Wd[d], pose_off[d]};
        }
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
        fold_frames = vgi::gram_needs_frames(p) && coupled.empty() && n_poses > 0 && !vgi::debug_hook(vgi::kHookSolverNoFoldFrames);
        for (int d = 0; d < n_ds && fold_frames; d++)
            if (vgi::gram_dataset_needs_frames(p, d) && seq_tf[d] < 0) fold_frames = false;
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
        x_cur = d_x.p;
        x_cand = d_xc.p;
        VG_TRY(d_delta.alloc((size_t)n_params));
        VG_TRY(d_rec.alloc((size_t)n_poses * vg::kPoseRec));
        VG_TRY(d_rows.alloc((size_t)n_rows * C));
        VG_TRY(d_rgroups.alloc((size_t)(n_groups > sg_wgs ? n_groups : sg_wgs) * (C * C + 1)));  // fused rows + Gram: C * C + 1 per workgroup
        // [Gram of the pose rows (C x C) | number of pose blocks that were not positive definite]: ONE buffer, so that the
        // count is summed over ranks by the same all-reduce and every rank takes the same accept / reject branch
        VG_TRY(d_rgram.alloc((size_t)C * C + 1));
        d_bad = d_rgram.p + (size_t)C * C;
        VG_TRY(d_dg.alloc((size_t)(G ? G : 1)));
        n_bs_groups = (unsigned int)((n_poses + vg::kBsPosesPerBlock - 1) / vg::kBsPosesPerBlock);
        // d_sums = [per-dataset summed Gram blocks (n_ds x Wmax^2) | scalar sums of the step (5)]: everything that is SUMMED
        // over ranks, contiguous, so that one evaluation ends with ONE in-place RCCL all-reduce of this buffer on the
        // problem's stream and one D2H.  d_small = [max |g_pose| (bit pattern) 1 | current global values G].
        n_sums = (size_t)n_ds * Wmax * Wmax;
        n_pack = n_sums + 5;
        VG_TRY(d_scal.alloc((size_t)n_bs_groups * 5));
        VG_TRY(d_small.alloc((size_t)1 + (size_t)(G ? G : 1)));
        // what is cleared / copied before the first evaluation: collected here, done by ONE launch at the head of the loop that runs
        std::memset(&init.h0, 0, sizeof init.h0);
        init.add_zero(d_small.p, 1 + (size_t)(G ? G : 1));
        init.add_zero(d_sums.p, n_pack);
        d_gmax = reinterpret_cast<unsigned long long *>(d_small.p);
        d_xg = d_small.p + 1;
        init.add_zero(d_delta.p, (size_t)(n_params ? n_params : 1));
        init.src = p->d_params;
        init.dst0 = d_x.p;
        init.n_copy = (unsigned long long)n_params;

        h_sums.assign((size_t)n_ds * Wmax * Wmax + 5, 0.);
        h_rgram.assign((size_t)C * C + 1, 0.);
        U.assign((size_t)G * G, 0.);
        gg.assign((size_t)G, 0.);
        Uc.assign((size_t)G * G, 0.);
        ggc.assign((size_t)G, 0.);
        S.assign((size_t)G * G, 0.);
        rhs.assign((size_t)G, 0.);
        dg.assign((size_t)G, 0.);
        h_xg.assign((size_t)G, 0.);
        // pin_small: [dg (G) | gmax (1) | xg (G)]
        VG_TRY(pin_sums.alloc(h_sums.size()));
        VG_TRY(pin_rgram.alloc(h_rgram.size()));
        VG_TRY(pin_small.alloc((size_t)2 * G + 2));
        // where the sum kernels deliver [summed Gram blocks | 5 step scalars]: the device buffer (all-reduced / read by the accept
        // kernel), or straight into the pinned block the host reads
        sums_out = host_direct ? pin_sums.p : d_sums.p;
        if (host_direct) {
            std::memset(pin_sums.p, 0, sizeof(double) * h_sums.size());
            std::memset(pin_small.p, 0, sizeof(double) * ((size_t)2 * G + 2));
        }
        d_scal_sum = sums_out + n_sums;
        // The host-driven loop on one rank does not wait through the runtime for the two read-backs of an iteration: a sequence
        // number lands in pinned memory and the host spins on it -- hipStreamSynchronize costs ~3.5 us more per round trip
        // (tools/exp/host_wait.hip).  The strided sum of the pose elimination (133 workgroups at G = 45) stores it when its last
        // workgroup is done (vg::HostSignal); behind an evaluation, whose last launch has 800 workgroups at the rig's size (an atomic
        // each cost more than the wait saves), a one-thread kernel does.  Rig, same box: 0.163-0.170 -> 0.157-0.158 ms per iteration.
        // vg_debug_set("solver_event_wait", 1): the runtime's wait (A/B).
        host_spin = host_direct && !vgi::debug_hook(vgi::kHookSolverEventWait);
        if (host_spin) {
            VG_TRY(pin_seq.alloc(2));
            VG_TRY(d_sigcnt.alloc(2));
            init.add_zero(reinterpret_cast<double *>(d_sigcnt.p), 1);   // two 32-bit counters
            std::memset(pin_seq.p, 0, sizeof(double) * 2);
        }
        mark("scratch + pinned allocation");
        return VG_OK;
    }

    // the descriptor tables of the fixed-order sums of the Gram blocks (several datasets: ONE slab launch and ONE final launch
    // for the two alternating Gram sets, or ONE launch over the per-workgroup partials of the merged vector-pipe launch)
    int setup_sum_tables()
    {
        int rc;
        sum_partials = std::vector<DevBuf<double>>((size_t)n_ds);
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
        use_partials = n_ds > 1 && !(opt.soft_l1_scale > 0.) && vgi::gram_merge_covers_all(p);
        wg_partials = std::vector<DevBuf<double>>((size_t)n_ds);
        wg_partials_ptr.assign((size_t)n_ds, nullptr);
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
        return VG_OK;
    }

    // clears, the starting point into the parameter buffer(s), the initial trust-region state: ONE launch
    int launch_init()
    {
        unsigned long long n_max = init.n_copy;
        for (int k = 0; k < vg::SolverInitArgs::kZero; k++) n_max = init.n_zero[k] > n_max ? init.n_zero[k] : n_max;
        const unsigned int grid = (unsigned int)std::min<unsigned long long>(std::max<unsigned long long>((n_max + 255) / 256, 1ull), 1024ull);
        hipLaunchKernelGGL(vg::vg_solver_init_kernel, dim3(grid), dim3(256), 0, st, init);
        VG_HIP(hipGetLastError());
        return VG_OK;
    }

    // ================================================================================================ waits of the host-driven loop
    vg::HostSignal host_signal(int which)   // 0: evaluation, 1: pose elimination
    {
        vg::HostSignal h;
        h.counter = d_sigcnt.p + which;
        h.host_seq = reinterpret_cast<unsigned long long *>(pin_seq.p) + which;
        h.seq = ++seq_issued[which];
        return h;
    }

    // Spin on a pinned word until it holds `expect`.  The first ~20 000 polls are bare (the answer is microseconds away and the
    // spin is the point: hipStreamSynchronize costs 3.5 us more per round trip); after that the thread pauses between polls so
    // that N in-process ranks on a CPU-quota'd box do not starve the rank that launches, and once a millisecond has passed it
    // asks the runtime: a stream that drained without the word means a failed launch -- reported then, not after 30 s.
    int spin_until(const volatile unsigned long long *w, unsigned long long expect, const char *what)
    {
        unsigned long spins = 0;
        double t_spin = 0.;
        while (*w != expect) {
            ++spins;
            if (spins < 20000) continue;
            __builtin_ia32_pause();
            if ((spins & 0x3ff) != 0) continue;
            const double t = now_s();
            if (t_spin == 0.) t_spin = t;
            if (t - t_spin < 1e-3) continue;
            const hipError_t q = hipStreamQuery(st);
            if (q == hipErrorNotReady) {
                if (t - t_spin > 30.) {   // the device is gone: do not hang
                    VG_HIP(hipStreamSynchronize(st));
                    if (*w != expect) return fail(VG_ERR_STATE, what);
                }
                continue;
            }
            VG_HIP(q);
            if (*w != expect) return fail(VG_ERR_STATE, what);   // everything queued has run and nobody reported
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        return VG_OK;
    }

    int host_wait(int which)
    {
        return spin_until(reinterpret_cast<volatile unsigned long long *>(pin_seq.p) + which, seq_issued[which],
                          "a sum kernel of an LM iteration never reported");
    }

    // ================================================================================================ evaluation
    // queue the evaluation of the Gram matrices at a device parameter buffer into gram set `set`, their fixed-order sums
    // into d_sums and the ONE collective of an evaluation (no host synchronisation)
    // `step_scalars`: the five scalar sums of the step that led to x_dev (+ its max |g_pose| to `gmax_out`) are wanted with this
    // evaluation: added by two more workgroups of the partial-sum launch when there is one, by vg_step_scalars_kernel otherwise
    int enqueue_evaluate(const double *x_dev, DevBuf<double> *set, bool frames_ready = false, bool step_scalars = false,
                         unsigned long long *gmax_out = nullptr)
    {
        int r;
        if (step_scalars && !use_partials) {
            hipLaunchKernelGGL(vg::vg_step_scalars_kernel, dim3(2), dim3(256), 0, st, (const double *)d_scal.p, n_bs_groups, d_scal_sum,
                               (const unsigned long long *)d_gmax, gmax_out);
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
                ssa.out = d_scal_sum;
                ssa.gmax_bits = d_gmax;
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
    }

    // evaluate at a device parameter buffer and assemble U / gg / cost on the host
    int evaluate(const double *x_dev, DevBuf<double> *set, std::vector<double> &Uo, std::vector<double> &go, double &cost2,
                 bool frames_ready = false, bool step_scalars = false, unsigned long long *gmax_out = nullptr)
    {
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
    }

    // sum a packed host buffer over ranks (multi-GPU); identity on one GPU
    int allreduce(std::vector<double> &buf)
    {
        if (!opt.allreduce) return VG_OK;
        return opt.allreduce(buf.data(), (int64_t)buf.size(), opt.allreduce_user) == 0 ? VG_OK : fail(VG_ERR_STATE, "allreduce callback failed");
    }

    // TransformationPrior blocks live on the host: r = A [R e_t; R e_r], e = prior^-1 o xi, Jacobian = A
    // (calib_cost_functions.cpp:214-228).  Added AFTER the all-reduce, identically on every rank.
    void add_priors(const std::vector<double> &xg_vals, std::vector<double> &Uo, std::vector<double> &go, double &c2)
    {
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
    }

    // the arguments of the pose elimination at Gram set `ds` (both loops)
    vg::SchurArgs schur_args(vg::SolveDatasetDev *ds) const
    {
        vg::SchurArgs sa;
        sa.mu = 0.;            // the caller's: the damping parameter by value (host loop) or where it lives on the device
        sa.mu_dev = nullptr;
        sa.gate = nullptr;
        sa.gate_expect = 0;
        sa.ds = ds;
        sa.inv = d_inv.p;
        sa.ref_ptr = d_ref_ptr.p;
        sa.ref_ds = d_ref_ds.p;
        sa.ref_blk = d_ref_blk.p;
        sa.pose_frozen = d_pf.p;
        sa.n_ds = n_ds;
        sa.G = G;
        sa.n_poses = (int)n_poses;
        sa.dmin = opt.min_lm_diagonal;
        sa.dmax = opt.max_lm_diagonal;
        sa.rec = d_rec.p;
        sa.rows = d_rows.p;
        sa.bad = d_bad;
        return sa;
    }

    // the solution into the problem's parameter vector, the summary
    int finish(int iterations, int n_success, int term, double initial_cost, double final_cost, double grad_max, double radius,
               const char *msg, bool device_resident, double t_loop)
    {
        VG_HIP(hipMemcpyAsync(p->d_params, x_cur, sizeof(double) * (size_t)n_params, hipMemcpyDeviceToDevice, st));
        VG_HIP(hipStreamSynchronize(st));
        p->frames_stale = true;  // whatever frames are in HBM belong to some candidate point, not to the solution
        if (!sum) return VG_OK;
        std::memset(sum, 0, sizeof *sum);
        sum->initial_cost = initial_cost;
        sum->final_cost = final_cost;
        sum->num_iterations = iterations;
        sum->num_successful_steps = n_success;
        sum->termination = term;
        sum->gradient_max_norm = grad_max;
        sum->final_radius = radius;
        sum->total_seconds = now_s() - t_start;
        if (device_resident) {
            sum->host_seconds = t_loop - t_start;            // set-up: buffers, index tables, uploads
            sum->evaluate_seconds = now_s() - t_loop;        // the iterations (device resident)
        } else {
            sum->evaluate_seconds = t_eval;
            sum->schur_seconds = t_schur;
            sum->host_seconds = t_host;
        }
        sum->num_global_columns = G;
        sum->num_pose_blocks = n_poses;
        std::snprintf(sum->message, sizeof sum->message, "%s", msg);
        return VG_OK;
    }

    // ================================================================================================ the two loops
    struct DeviceLoop;   // vg_lm_device_loop.hpp
    struct HostLoop;     // vg_lm_host_loop.hpp
    int run_device_loop();
    int run_host_loop();
};

}  // namespace
