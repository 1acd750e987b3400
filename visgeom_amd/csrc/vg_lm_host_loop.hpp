// vg_lm_host_loop.hpp -- LmSolve::HostLoop: the host-driven Levenberg-Marquardt loop (see vg_lm_solve.hpp): wide reduced
// systems (the rig: the host factorises 45 x 45 in ~3 us behind a 9 us round trip where one workgroup needs 22 us), priors,
// odometry-coupled sequences (their block-tridiagonal elimination runs on the host), the all-reduce callback.  Ceres' trust-region
// policy (acceptance, radius, the three convergence tests) as in the device-resident loop's accept kernel.
// Per iteration: eliminate_poses (pose rows + their Gram on the device, Schur complement to the host) -> reduced_solve (damped
// G x G system with the active set of the box bounds, host Cholesky) -> step_and_evaluate (back-substitution + the candidate's
// frames + evaluation of the candidate, the step's scalar sums, the convergence tests) -> accept or shrink.
#pragma once

namespace {

struct LmSolve::HostLoop {
    LmSolve &s;
    DevBuf<double> *cur = nullptr, *cand = nullptr;            // Gram set of the current / candidate point
    vg::SolveDatasetDev *ds_cur = nullptr, *ds_cand = nullptr;
    std::vector<double> h_xcur;    // global values at the CURRENT point (s.h_xg is refreshed only after the reduced solve)
    double cost2 = 0., cost2_c = 0., radius = 0., decrease_factor = 2., grad_max = 0., initial_cost = 0.;
    int n_success = 0, term = VG_TERM_NO_CONVERGENCE;
    char msg[160] = "";
    std::vector<unsigned char> held;
    std::vector<double> Sw, rw, chol_ws;
    // one iteration's numbers
    struct Step {
        double mu = 0., model_change = 0., step2 = 0., cost_change = 0., rho = 0.;
        bool coupled_ok = true, step_ok = false, stop = false;   // stop: a convergence test fired (term / msg are set)
        const double *rg = nullptr;                               // the Schur complement [C x C | count of bad pose blocks] as the host reads it
    };

    explicit HostLoop(LmSolve &solve) : s(solve) {}

    // uploads, the init launch, the global values and the evaluation at the starting point
    int start()
    {
        int rc;
        hipStream_t st = s.st;
        const int G = s.G;
        if (t_arena) VG_TRY(t_arena->flush(st));
        VG_TRY(s.launch_init());
        // values of the global columns at the starting point
        // (ONE copy of the span they lie in -- the global blocks are neighbours in the parameter vector -- not a blocking copy per
        // column: 45 x 20 us in front of the rig's first iteration, rocprofv3 trace)
        if (G) {
            long long lo_p = s.gcol_param[0], hi_p = s.gcol_param[0];
            for (int a2 = 1; a2 < G; a2++) {
                lo_p = s.gcol_param[a2] < lo_p ? s.gcol_param[a2] : lo_p;
                hi_p = s.gcol_param[a2] > hi_p ? s.gcol_param[a2] : hi_p;
            }
            std::vector<double> span((size_t)(hi_p - lo_p + 1));
            VG_HIP(hipMemcpyAsync(span.data(), s.p->d_params + lo_p, sizeof(double) * span.size(), hipMemcpyDeviceToHost, st));
            VG_HIP(hipStreamSynchronize(st));
            for (int a2 = 0; a2 < G; a2++) s.h_xg[a2] = span[(size_t)(s.gcol_param[a2] - lo_p)];
        }
        h_xcur = s.h_xg;
        cur = s.gramA;
        cand = s.gramB;
        ds_cur = s.d_dsA.p;
        ds_cand = s.d_dsB.p;
        VG_TRY(s.evaluate(s.x_cur, cur, s.U, s.gg, cost2));
        {
            std::vector<double> pack(s.U);
            pack.insert(pack.end(), s.gg.begin(), s.gg.end());
            pack.push_back(cost2);
            VG_TRY(s.allreduce(pack));
            std::copy(pack.begin(), pack.begin() + (size_t)G * G, s.U.begin());
            std::copy(pack.begin() + (size_t)G * G, pack.begin() + (size_t)G * G + G, s.gg.begin());
            cost2 = pack.back();
            s.add_priors(s.h_xg, s.U, s.gg, cost2);
        }
        for (auto &c2 : s.coupled) {
            VG_HIP(hipMemcpy(c2.x.data(), s.x_cur + c2.param_off, sizeof(double) * c2.x.size(), hipMemcpyDeviceToHost));
            cost2 += c2.cost2(c2.x, s.h_xg.data());
            c2.add_global_terms(c2.x, s.h_xg.data(), G, s.U, s.gg);
        }
        radius = s.opt.initial_trust_region_radius;
        decrease_factor = 2.;
        initial_cost = 0.5 * cost2;
        if (s.opt.verbose) std::printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n%4d  %.6e\n", 0, initial_cost);
        if (!std::isfinite(cost2)) {  // as Ceres: a failed evaluation of the starting point fails the solve
            term = VG_TERM_FAILURE;
            std::snprintf(msg, sizeof msg, "the cost at the starting point is not finite (NaN / Inf in the residuals)");
        }
        return VG_OK;
    }

    // ---- eliminate the poses: rows -> Gram -> S_sub, c; the Schur complement of all ranks where the host reads it (q.rg)
    int eliminate_poses(vg::SchurArgs &sa, Step &q)
    {
        int rc;
        hipStream_t st = s.st;
        const int G = s.G, C = s.C;
        const vg_solve_options &opt = s.opt;
        const vg_comm *comm = s.comm;
        const double t0 = now_s();
        sa = s.schur_args(ds_cur);
        sa.mu = q.mu;
        sa.mu_dev = nullptr;
        sa.gate = nullptr;
        sa.gate_expect = 0;
        // the Schur complement is read where the device wrote it when this rank's kernels deliver it straight to pinned memory;
        // otherwise (all-reduce callback, no poses) from the staging vector
        const bool rgram_in_place = s.host_direct && s.n_poses > 0;
        if (!rgram_in_place) std::fill(s.h_rgram.begin(), s.h_rgram.end(), 0.);
        q.rg = rgram_in_place ? s.pin_rgram.p : s.h_rgram.data();
        const bool schur_spin = s.host_spin && s.n_poses > 0 && s.coupled.empty();
        q.coupled_ok = true;
        if (s.n_poses) {
            if (s.coupled.empty()) {
                sa.zero_u64 = s.d_gmax;  // the step's max |g_pose|, cleared here instead of by a memset in front of the back-substitution
                hipLaunchKernelGGL(vg::vg_schur_rows_gram_kernel, dim3(s.sg_wgs), dim3(vg::kSchurThreads * s.sg_batches), s.sg_lds, st, sa, s.sg_ppw, s.sg_batches, s.d_rgroups.p, s.sg_shared);
            } else {
                VG_HIP(hipMemsetAsync(s.d_bad, 0, sizeof(double), st));
                hipLaunchKernelGGL(vg::vg_schur_rows_kernel, dim3((unsigned)((s.n_poses * C + 255) / 256)), dim3(256), 0, st, sa);
            }
            VG_HIP(hipGetLastError());
            // sequences coupled by odometry: raw V / g / W^T come back, the host eliminates the block-tridiagonal
            // system and puts its rows where the per-pose rows would be
            for (auto &c2 : s.coupled) {
                std::vector<double> hrec((size_t)c2.n * vg::kPoseRec), hraw((size_t)c2.n * 6 * C);
                if (s.coupled_multi) {  // raw normal-equation pieces of the replicated sequence, summed over the ranks' images
                    VG_TRY(vgc::allreduce_sum(comm, s.d_rec.p + (size_t)c2.pb * vg::kPoseRec, hrec.size(), st));
                    VG_TRY(vgc::allreduce_sum(comm, s.d_rows.p + (size_t)c2.pb * 6 * C, hraw.size(), st));
                }
                VG_HIP(hipMemcpyAsync(hrec.data(), s.d_rec.p + (size_t)c2.pb * vg::kPoseRec, sizeof(double) * hrec.size(),
                                      hipMemcpyDeviceToHost, st));
                VG_HIP(hipMemcpyAsync(hraw.data(), s.d_rows.p + (size_t)c2.pb * 6 * C, sizeof(double) * hraw.size(),
                                      hipMemcpyDeviceToHost, st));
                VG_HIP(hipStreamSynchronize(st));
                if (!c2.eliminate(hrec.data(), hraw.data(), G, q.mu, opt.min_lm_diagonal, opt.max_lm_diagonal, h_xcur.data())) {
                    q.coupled_ok = false;
                    std::fill(c2.Y.begin(), c2.Y.end(), 0.);
                    c2.Y.resize((size_t)c2.n * 6 * C, 0.);
                }
                // the rows enter the Schur complement ONCE: every rank has the same ones, rank 0 contributes them
                if (s.coupled_multi && comm->rank != 0) {
                    VG_HIP(hipMemsetAsync(s.d_rows.p + (size_t)c2.pb * 6 * C, 0, sizeof(double) * c2.Y.size(), st));
                } else {
                    VG_HIP(hipMemcpyAsync(s.d_rows.p + (size_t)c2.pb * 6 * C, c2.Y.data(), sizeof(double) * c2.Y.size(),
                                          hipMemcpyHostToDevice, st));
                }
                VG_HIP(hipStreamSynchronize(st));  // c2.Y may be rewritten before an async copy from pageable memory ends
            }
            if (s.coupled.empty()) {
                vg::launch_strided_sum(st, s.d_rgroups.p, s.sg_wgs, C * C + 1, s.host_direct ? s.pin_rgram.p : s.d_rgram.p,
                                       schur_spin ? s.host_signal(1) : vg::HostSignal());
            } else {
                VG_TRY(launch_dense_gram(st, s.d_rows.p, s.n_rows, C, s.rows_per_group, s.n_groups, s.d_rgroups.p));
                vg::launch_strided_sum(st, s.d_rgroups.p, s.n_groups, C * C, s.d_rgram.p);
            }
            VG_HIP(hipGetLastError());
        } else if (comm && comm->n_ranks > 1) {
            VG_HIP(hipMemsetAsync(s.d_rgram.p, 0, sizeof(double) * s.h_rgram.size(), st));  // a rank without poses still joins the sum
        }
        if (s.n_poses || (comm && comm->n_ranks > 1)) {
            if (!s.host_direct) {
                VG_TRY(vgc::allreduce_sum(comm, s.d_rgram.p, s.h_rgram.size(), st));  // Schur complement of the poses of all ranks
                VG_HIP(hipMemcpyAsync(s.pin_rgram.p, s.d_rgram.p, sizeof(double) * s.h_rgram.size(), hipMemcpyDeviceToHost, st));
            }
            if (schur_spin) VG_TRY(s.host_wait(1));
            else VG_HIP(hipStreamSynchronize(st));
            if (!rgram_in_place) std::memcpy(s.h_rgram.data(), s.pin_rgram.p, sizeof(double) * s.h_rgram.size());
        }
        if (opt.allreduce) VG_TRY(s.allreduce(s.h_rgram));   // (host_direct excludes the callback: rg stays valid)
        // poses whose damped 6 x 6 block was not positive definite (NaN / Inf in their Gram block): the step is invalid
        // as a whole -- rejected like a failed factorisation of the reduced system, and counted.  The count is the one
        // summed over ALL ranks (last slot of the buffer): a rank-local decision here would make this rank skip the
        // collectives of the candidate evaluation while the others enter them.
        if (q.rg[(size_t)C * C] > 0.) {
            q.coupled_ok = false;
            s.n_bad_pose_blocks += (long long)q.rg[(size_t)C * C];
        }
        s.t_schur += now_s() - t0;
        return VG_OK;
    }

    // ---- reduced system on the host: S = U - Y^T Y + mu D, rhs = -g + Y^T c, constant blocks and the active set of the bounds
    void reduced_solve(Step &q)
    {
        const int G = s.G, C = s.C;
        const vg_solve_options &opt = s.opt;
        const double t0 = now_s();
        const double *rg = q.rg;
        std::vector<double> &S = s.S, &U = s.U, &rhs = s.rhs, &dg = s.dg;
        // (rows 0 .. G - 1 of the Schur complement's lower triangle and its last ROW, which is its last column: half the cache
        //  lines of what the device wrote)
        for (int a2 = 0; a2 < G; a2++) {
            for (int b2 = 0; b2 <= a2; b2++) {
                const double r2 = rg[(size_t)a2 * C + b2];
                S[(size_t)a2 * G + b2] = U[(size_t)a2 * G + b2] - r2;
                if (a2 != b2) S[(size_t)b2 * G + a2] = U[(size_t)b2 * G + a2] - r2;
            }
            const double dd = U[(size_t)a2 * G + a2];
            S[(size_t)a2 * G + a2] += q.mu * (dd < opt.min_lm_diagonal ? opt.min_lm_diagonal : (dd > opt.max_lm_diagonal ? opt.max_lm_diagonal : dd));
            rhs[a2] = -s.gg[a2] + rg[(size_t)G * C + a2];
        }
        // Constant blocks, and the active set of the box bounds: a parameter sitting ON a bound whose step points
        // outwards is held for this iteration (its row / column leave the reduced system -- the Schur complement of
        // the constrained problem is exactly that sub-matrix).  Without this the projected step keeps "spending" its
        // decrease on a coordinate that cannot move, the gain ratio collapses and the radius shrinks to nothing.
        held.assign(s.gfrozen.begin(), s.gfrozen.end());   // (held, Sw, rw, chol_ws: allocated once, kept over the iterations)
        bool step_ok = q.coupled_ok;
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
                    const double l2 = s.glo[(size_t)a2], h2 = s.ghi[(size_t)a2];
                    if ((h_xcur[a2] <= l2 && dg[a2] < 0.) || (h_xcur[a2] >= h2 && dg[a2] > 0.)) {
                        held[a2] = 1;
                        changed = true;
                    }
                }
            if (!changed) break;
        }
        q.step_ok = step_ok;
        s.t_host += now_s() - t0;
    }

    // ---- back-substitute, apply, evaluate the candidate; the step's scalar sums; gradient / parameter / function tolerance
    int step_and_evaluate(const vg::SchurArgs &sa, Step &q)
    {
        int rc;
        hipStream_t st = s.st;
        const int G = s.G;
        const vg_solve_options &opt = s.opt;
        const vg_comm *comm = s.comm;
        std::vector<double> &dg = s.dg, &h_xg = s.h_xg, &Uc = s.Uc, &ggc = s.ggc;
        const bool host_direct = s.host_direct;
        double t0 = now_s();
        if (G) {
            std::memcpy(s.pin_small.p, dg.data(), sizeof(double) * G);
            if (!host_direct) VG_HIP(hipMemcpyAsync(s.d_dg.p, s.pin_small.p, sizeof(double) * G, hipMemcpyHostToDevice, st));
        }
        if (!(s.n_poses && s.coupled.empty())) VG_HIP(hipMemsetAsync(s.d_gmax, 0, sizeof(unsigned long long), st));  // else: cleared by the rows kernel
        vg::BacksubArgs ba;
        ba.s = sa;
        ba.dg = host_direct ? s.pin_small.p : s.d_dg.p;   // the reduced step: read where the host wrote it
        ba.pose_param = s.d_pose_param.p;
        ba.gcol_param = s.d_gcol_param.p;
        ba.delta = s.d_delta.p;
        ba.scal = s.d_scal.p;
        ba.gmax_bits = s.d_gmax;
        ba.x = s.x_cur;
        double *ps = s.pin_small.p + G;  // [gmax 1 | xg G]
        ba.xg = host_direct ? ps + 1 : s.d_xg;   // current values of the global columns, for the host
        ba.lo = s.d_glo.p;
        ba.hi = s.d_ghi.p;
        ba.x_new = s.x_cand;   // host-eliminated sequences overwrite their poses below
        ba.fold = s.fold_frames ? s.d_fold.p : nullptr;   // the candidate's frames come out of the same launch
        ba.fold_gcol = s.d_fold_gcol.p;
        if (s.n_poses || G) {  // G <= kBsThreads: one workgroup is enough for the global columns alone
            const unsigned int bs_grid = s.n_bs_groups ? s.n_bs_groups : 1u;
            vg::launch_backsub(st, G, bs_grid, ba);
            VG_HIP(hipGetLastError());
        }
        // (the fixed-order sum of the back-substitution's per-workgroup partials and, host_direct, max |g_pose| to the host:
        //  with the candidate's evaluation below)
        double host_scal[5] = {0., 0., 0., 0., 0.};
        for (auto &c2 : s.coupled) {
            std::vector<double> dp;
            double sc[5];
            c2.backsub(dg.data(), G, dp, sc);
            for (int k = 0; k < 4; k++) host_scal[k] += sc[k];
            host_scal[4] = sc[4] > host_scal[4] ? sc[4] : host_scal[4];
            VG_HIP(hipMemcpyAsync(s.d_delta.p + c2.param_off, dp.data(), sizeof(double) * dp.size(), hipMemcpyHostToDevice, st));
            VG_HIP(hipStreamSynchronize(st));
        }
        // the back-substitution kernel wrote the candidate of every global column and of every pose it owns; the
        // poses of host-eliminated sequences (unbounded) take their steps here
        for (auto &c2 : s.coupled) {
            const long long n2 = (long long)c2.n * 6;
            hipLaunchKernelGGL(vg::vg_apply_step_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st,
                               (const double *)s.x_cur + c2.param_off, (const double *)s.d_delta.p + c2.param_off, n2,
                               s.x_cand + c2.param_off);
            VG_HIP(hipGetLastError());
        }
        // without poses the five scalar sums (tail of the sums block) stay at the zeros they were initialised with, and so
        // does max |g_pose|
        if (!host_direct) VG_HIP(hipMemcpyAsync(ps, s.d_small.p, sizeof(double) * (1 + (size_t)G), hipMemcpyDeviceToHost, st));
        s.t_schur += now_s() - t0;
        // No wait here: the candidate evaluation does not depend on these scalars, it is queued right behind the
        // step on the same stream, and its own read-back synchronises once for both (one host round trip per
        // iteration less; the wait is booked under "evaluate").
        VG_TRY(s.evaluate(s.x_cand, cand, Uc, ggc, cost2_c, s.fold_frames && s.n_poses > 0, s.n_bs_groups > 0,
                          host_direct ? reinterpret_cast<unsigned long long *>(ps) : nullptr));

        // |x|^2 of this rank's pose parameters (summed over ranks below) and of the replicated global block
        for (int a2 = 0; a2 < G; a2++) h_xg[a2] = ps[1 + a2];
        const double *sc = s.pin_sums.p + s.n_sums;  // scalar sums of the step, already summed over ranks with an RCCL communicator
        double xg2 = 0.;
        for (int a2 = 0; a2 < G; a2++) xg2 += h_xg[a2] * h_xg[a2];
        double gdp = sc[0] + host_scal[0], ddp = sc[1] + host_scal[1], dp2 = sc[2] + host_scal[2],
               gp2 = sc[3] + host_scal[3], xp2 = sc[4], gmax_p = ps[0] > host_scal[4] ? ps[0] : host_scal[4];
        // global values of the candidate: clamp(x + dg), as vg_apply_step_kernel does
        std::vector<double> xg_c(G);
        for (int a2 = 0; a2 < G; a2++) {
            const double v = h_xg[a2] + dg[a2], l2 = s.glo[(size_t)a2], h2 = s.ghi[(size_t)a2];
            xg_c[a2] = v < l2 ? l2 : (v > h2 ? h2 : v);
        }
        for (auto &c2 : s.coupled) {
            VG_HIP(hipMemcpy(c2.xc.data(), s.x_cand + c2.param_off, sizeof(double) * c2.xc.size(), hipMemcpyDeviceToHost));
            cost2_c += c2.cost2(c2.xc, xg_c.data());
            if (s.coupled_multi) {  // the replicated poses entered the summed |x|^2 once per rank
                double x2 = 0.;
                for (double v : c2.x) x2 += v * v;
                xp2 -= (double)(comm->n_ranks - 1) * x2;
            }
        }
        if (opt.allreduce) {   // (no callback: nothing to pack, sum and unpack)
            std::vector<double> pack(Uc);
            pack.insert(pack.end(), ggc.begin(), ggc.end());
            pack.push_back(cost2_c);
            pack.push_back(gdp);
            pack.push_back(ddp);
            pack.push_back(dp2);
            pack.push_back(gp2);
            pack.push_back(xp2);
            VG_TRY(s.allreduce(pack));
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
        if (s.multi_rank) gmax_p = std::sqrt(gp2);
        if (!s.p->priors.empty()) s.add_priors(xg_c, Uc, ggc, cost2_c);
        for (auto &c2 : s.coupled) c2.add_global_terms(c2.xc, xg_c.data(), G, Uc, ggc);

        double gdg = 0., ddg = 0., dg2 = 0., gmax_g = 0.;
        for (int a2 = 0; a2 < G; a2++) {
            if (s.gfrozen[a2]) continue;
            const double dd = s.U[(size_t)a2 * G + a2];
            const double dcl = dd < opt.min_lm_diagonal ? opt.min_lm_diagonal : (dd > opt.max_lm_diagonal ? opt.max_lm_diagonal : dd);
            gdg += s.gg[a2] * dg[a2];
            ddg += dcl * dg[a2] * dg[a2];
            dg2 += dg[a2] * dg[a2];
            // projected gradient for bounded parameters: |Project(x - g) - x|
            const double xv = h_xg[a2];
            double xg = xv - s.gg[a2];
            const double l2 = s.glo[(size_t)a2], h2 = s.ghi[(size_t)a2];
            xg = xg < l2 ? l2 : (xg > h2 ? h2 : xg);
            gmax_g = std::fabs(xg - xv) > gmax_g ? std::fabs(xg - xv) : gmax_g;
        }
        grad_max = gmax_g > gmax_p ? gmax_g : gmax_p;
        // model decrease of the exact LM step: 1/2 delta^T (mu D delta - g)
        q.model_change = 0.5 * (q.mu * (ddg + ddp) - (gdg + gdp));
        q.step2 = dg2 + dp2;
        q.cost_change = 0.5 * (cost2 - cost2_c);
        q.rho = q.model_change > 0. ? q.cost_change / q.model_change : -1.;

        if (grad_max <= opt.gradient_tolerance) {
            term = VG_TERM_CONVERGENCE_GRADIENT;
            std::snprintf(msg, sizeof msg, "gradient tolerance reached: max norm %.3e <= %.3e", grad_max, opt.gradient_tolerance);
            q.stop = true;
            return VG_OK;
        }
        const double xn2 = xg2 + xp2;  // identical on every rank
        if (std::sqrt(q.step2) <= opt.parameter_tolerance * (std::sqrt(xn2) + opt.parameter_tolerance)) {
            term = VG_TERM_CONVERGENCE_PARAMETER;
            std::snprintf(msg, sizeof msg, "parameter tolerance reached: |step| %.3e", std::sqrt(q.step2));
            q.stop = true;
            return VG_OK;
        }
        // Function tolerance: Ceres tests |cost change| of EVERY evaluated candidate of a valid step, before it decides whether
        // the step is accepted (trust_region_minimizer.cc: the "function tolerance reached" block / FunctionToleranceReached()
        // sits in front of the relative-decrease test), and returns at the CURRENT point.  With the reference's 1e-15 this is
        // what ends the cascade of rejected noise-level steps at the tail of a solve after three or four radius reductions
        // instead of the eight the parameter tolerance needs; until round 4 the test ran for accepted steps only.
        if (q.model_change > 0. && std::isfinite(cost2_c) && std::fabs(cost2 - cost2_c) <= opt.function_tolerance * cost2) {
            term = VG_TERM_CONVERGENCE_FUNCTION;
            std::snprintf(msg, sizeof msg, "function tolerance reached: |cost change| / cost = %.3e",
                          cost2 > 0 ? std::fabs(cost2 - cost2_c) / cost2 : 0.);
            q.stop = true;
        }
        return VG_OK;
    }

    // Ceres' step acceptance and trust-region radius policy; returns false when the radius fell below its minimum
    bool accept_or_shrink(int iter, const Step &q)
    {
        const int G = s.G;
        const vg_solve_options &opt = s.opt;
        const bool success = q.step_ok && std::isfinite(cost2_c) && q.rho > opt.min_relative_decrease;
        if (opt.verbose)
            std::printf("%4d  %.6e  %10.3e  %10.3e  %9.3e  %9.3e  %9.3e %s\n", iter, 0.5 * (success ? cost2_c : cost2), q.cost_change,
                        grad_max, std::sqrt(q.step2), q.rho, radius, success ? "" : "(rejected)");
        if (success) {
            n_success++;
            std::swap(cur, cand);
            std::swap(ds_cur, ds_cand);
            std::swap(s.x_cur, s.x_cand);
            for (auto &c2 : s.coupled) c2.x.swap(c2.xc);
            for (int a2 = 0; a2 < G; a2++) {  // what vg_apply_step_kernel wrote: clamp(x + dg)
                const double v = s.h_xg[a2] + s.dg[a2], l2 = s.glo[(size_t)a2], h2 = s.ghi[(size_t)a2];
                h_xcur[a2] = v < l2 ? l2 : (v > h2 ? h2 : v);
            }
            s.U.swap(s.Uc);
            s.gg.swap(s.ggc);
            cost2 = cost2_c;
            const double f = 1. - std::pow(2. * q.rho - 1., 3);
            radius = radius / (f > 1. / 3. ? f : 1. / 3.);
            radius = radius > opt.max_trust_region_radius ? opt.max_trust_region_radius : radius;
            decrease_factor = 2.;
            return true;
        }
        radius /= decrease_factor;
        decrease_factor *= 2.;
        if (radius < opt.min_trust_region_radius) {
            term = VG_TERM_RADIUS_TOO_SMALL;
            std::snprintf(msg, sizeof msg, "trust region radius below %.1e", opt.min_trust_region_radius);
            return false;
        }
        return true;
    }

    int run()
    {
        int rc;
        const vg_solve_options &opt = s.opt;
        VG_TRY(start());
        int iter = 0;
        for (iter = 1; term != VG_TERM_FAILURE && iter <= opt.max_num_iterations; iter++) {
            Step q;
            q.mu = 1. / radius;
            vg::SchurArgs sa;
            VG_TRY(eliminate_poses(sa, q));
            reduced_solve(q);
            if (q.step_ok) {
                VG_TRY(step_and_evaluate(sa, q));
                if (q.stop) break;
            }
            if (!accept_or_shrink(iter, q)) break;
        }
        if (term == VG_TERM_FAILURE) iter = 0;
        else if (iter > opt.max_num_iterations) {
            iter = opt.max_num_iterations;
            std::snprintf(msg, sizeof msg, "maximum number of iterations reached");
        }
        if (s.n_bad_pose_blocks) {
            const size_t len = std::strlen(msg);
            std::snprintf(msg + len, sizeof msg - len, "%s%lld pose block(s) not positive definite", len ? "; " : "", s.n_bad_pose_blocks);
        }
        return s.finish(iter, n_success, term, initial_cost, 0.5 * cost2, grad_max, radius, msg, false, 0.);
    }
};

int LmSolve::run_host_loop()
{
    HostLoop loop(*this);
    return loop.run();
}

}  // namespace
