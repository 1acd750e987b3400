// vg_solver_host_loop.inc.hpp -- a FRAGMENT of vg_problem_solve (vg_solver_impl.hpp), included inside its body behind the
// device-resident loop: the host-driven Levenberg-Marquardt loop (wide reduced systems, priors, odometry-coupled sequences, the
// all-reduce callback) up to the summary.  It uses the function's locals (tables, buffers, the evaluate / launch_init lambdas,
// VG_TRY).  Split off for reading only.
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
