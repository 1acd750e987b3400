// vg_solver_device_loop.inc.hpp -- a FRAGMENT of vg_problem_solve (vg_solver_impl.hpp), included inside its body after the
// set-up: the device-resident Levenberg-Marquardt loop.  It uses the function's locals (the problem's tables and buffers, the
// enqueue_evaluate / launch_init lambdas, VG_TRY) and returns from the function when it ran.  Split off for reading only
// (VERDICT r3 next #9): the two loops share some eighty locals of the set-up, so they stay one function.
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
