// vg_lm_device_loop.hpp -- LmSolve::DeviceLoop: the device-resident Levenberg-Marquardt loop (see vg_lm_solve.hpp).
//
// Problems made of grid blocks only (no priors, no odometry, no host-staged all-reduce): the reduced solve and the step
// acceptance run on the device, the trust-region state lives there, and an iteration is a fixed sequence of six launches; the host
// only reads the state the accept kernel publishes in pinned memory.
// Measured (tools/exp/solve_probe.py, tools/prof_solve.py, one MI355X): 10 k EUCM images 0.083 ms per iteration, Mei 0.089; the
// 45-column rig 0.167 against 0.157 for the host-driven loop -- there the one-workgroup factorisation of the reduced system costs
// more than the host's round trip, so wide systems keep the host loop.  vg_debug_set("solver_host_loop" / "solver_device_loop")
// force a side.
//
// An iteration is a fixed sequence of launches whose buffers depend only on the PARITY of the number of accepted steps so far
// (which Gram set / parameter buffer is "current"); the device keeps that parity in LmState::gate.  So the host queues iteration
// k + 1 for the parity an acceptance of step k would give BEFORE it knows the outcome of step k -- every kernel of a queued
// iteration returns at once if the gate says otherwise (a rejected step: the same parity is queued again; convergence:
// gate = -1) -- and only then waits for the state of iteration k.  The GPU always has the next iteration in its queue: no launch
// latency, no idle time behind the host's read-back.  Robust (SoftLOne) evaluations re-weight the Gram set in place with an
// ungated kernel, so those solves queue one iteration at a time.  Several ranks speculate too: every rank holds the same state,
// so every rank queues the same launches and the same collectives; a collective of an iteration that skips itself is NOT
// skipped -- it runs on every rank, on buffers nobody reads (whatever a real iteration reads it has rewritten or cleared before
// its own collective).  Replaying the gated iteration as a hipGraph (one per parity) was slower than queueing its launches
// (profiles/NOTES.md).  vg_debug_set("solver_no_speculation", 1) queues one iteration at a time.
#pragma once

namespace {

struct LmSolve::DeviceLoop {
    LmSolve &s;
    static constexpr int kSlots = 4;
    DevBuf<vg::LmState> d_state;
    DevBuf<double> d_U, d_gvec, d_S, d_xcur;
    DevBuf<int> d_Wd;
    DevBuf<unsigned char> d_gfrozen;
    vg::LmAcceptArgs aa;
    vg::LmSolveArgs ra;
    size_t accept_lds = 0, solve_lds = 0;
    bool speculate = false, spin_wait = true;
    DevBuf<double> *gset[2] = {nullptr, nullptr};
    vg::SolveDatasetDev *dset[2] = {nullptr, nullptr};
    double *xbuf[2] = {nullptr, nullptr};
    // The accept kernel writes its state into a pinned slot itself and a per-slot SEQUENCE WORD behind it (system-scope release);
    // the host learns the outcome of an iteration by SPINNING on that word -- not from an event recorded behind the kernel: the
    // event's marker packet kept the next iteration's first kernel waiting 5-6 us after every accept (rocprofv3 trace,
    // tools/exp/trace_gaps.py).  vg_debug_set("solver_event_wait", 1) restores the event (A/B).
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
    unsigned long long seq_counter = 0ull;
    int n_queued = 0;

    explicit DeviceLoop(LmSolve &solve) : s(solve) {}

    int next_slot() { return n_queued++ % kSlots; }
    void arm_slot(int slot, vg::LmAcceptArgs &args)
    {
        args.host_state = slots.p + slot;
        if (spin_wait) {
            slots.expect[slot] = ++seq_counter;
            args.host_seq = const_cast<unsigned long long *>(slots.seq + slot);
            args.seq = slots.expect[slot];
        }
    }
    int queue_state(int slot)
    {
        if (!spin_wait) VG_HIP(hipEventRecord(slots.ev[slot], s.st));
        return VG_OK;
    }
    int wait_state(int slot)
    {
        if (!spin_wait) {
            VG_HIP(hipEventSynchronize(slots.ev[slot]));
            return VG_OK;
        }
        return s.spin_until(slots.seq + slot, slots.expect[slot], "the accept kernel of an LM iteration never reported");
    }

    // buffers of the loop, the arguments of its two one-workgroup kernels, the pinned slots
    int prepare()
    {
        int rc;
        const int G = s.G, n_ds = s.n_ds, Wmax = s.Wmax;
        const vg_solve_options &opt = s.opt;
        VG_TRY(d_state.alloc(1));
        VG_TRY(d_U.alloc((size_t)2 * G * G));
        VG_TRY(d_gvec.alloc((size_t)2 * G));
        VG_TRY(d_S.alloc((size_t)G * G));
        VG_TRY(d_xcur.alloc((size_t)G));
        VG_TRY(d_Wd.upload(s.Wd));
        VG_TRY(d_gfrozen.upload(s.gfrozen));
        vg::LmState &h0 = s.init.h0;
        h0.radius = opt.initial_trust_region_radius;
        h0.decrease_factor = 2.;
        h0.mu = 1. / h0.radius;
        h0.term = VG_TERM_NO_CONVERGENCE;
        s.init.state = d_state.p;
        s.init.add_zero(s.d_rgram.p, s.h_rgram.size());  // also the bad-pose counter behind it
        s.init.dst1 = s.d_xc.p;

        aa.st = d_state.p;
        aa.U = d_U.p;
        aa.gg = d_gvec.p;
        aa.sums = s.d_sums.p;
        aa.inv = s.d_inv.p;
        aa.Wd = d_Wd.p;
        aa.dg = s.d_dg.p;
        aa.gmax_bits = s.d_gmax;
        aa.bad = s.d_bad;
        aa.xcur = d_xcur.p;
        aa.x = s.d_x.p;
        aa.gcol_param = s.d_gcol_param.p;
        aa.lo = s.d_glo.p;
        aa.hi = s.d_ghi.p;
        aa.gfrozen = d_gfrozen.p;
        aa.n_ds = n_ds;
        aa.Wmax = Wmax;
        aa.G = G;
        aa.init = 1;
        aa.multi_rank = s.multi_rank ? 1 : 0;
        aa.scal_partials = (s.n_bs_groups && !s.multi_rank) ? s.d_scal.p : nullptr;
        aa.n_scal = s.n_bs_groups;
        accept_lds = sizeof(double) * ((size_t)n_ds * Wmax * Wmax + ((size_t)n_ds * G + 1) / 2 + 1);
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
        ra.st = d_state.p;
        ra.U = d_U.p;
        ra.gg = d_gvec.p;
        ra.rgram = s.d_rgram.p;
        ra.lo = s.d_glo.p;
        ra.hi = s.d_ghi.p;
        ra.gfrozen = d_gfrozen.p;
        ra.xcur = d_xcur.p;
        ra.dg = s.d_dg.p;
        ra.S = d_S.p;
        ra.G = G;
        ra.use_bounds = opt.use_bounds;
        ra.dmin = opt.min_lm_diagonal;
        ra.dmax = opt.max_lm_diagonal;
        const bool s_in_lds = sizeof(double) * (2 * (size_t)G * G + 4 * (size_t)G + 2) <= 150 * 1024;
        if (s_in_lds) ra.S = nullptr;
        solve_lds = sizeof(double) * ((s_in_lds ? 2 : 1) * (size_t)G * G + 4 * (size_t)G + 2);
        if (solve_lds > 64 * 1024)  // up to 127 global columns: 133 KB of the CU's 160 KB
            VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vg::vg_lm_reduced_solve_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_lds));
        if (G <= vg::kEntrySolveMaxG && sizeof(double) * vg::lm_entry_solve_lds_doubles(G) > 48 * 1024)   // 51 KB at G = 63
            VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vg::vg_lm_reduced_solve_entries_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * vg::lm_entry_solve_lds_doubles(G))));
        aa.gate_expect = -1;
        ra.gate_expect = -1;
        speculate = opt.soft_l1_scale <= 0. && vgi::debug_hook(vgi::kHookSolverNoSpeculation) != 1;
        gset[0] = s.gramA;
        gset[1] = s.gramB;
        dset[0] = s.d_dsA.p;
        dset[1] = s.d_dsB.p;
        xbuf[0] = s.d_x.p;
        xbuf[1] = s.d_xc.p;
        const size_t slots_bytes = sizeof(vg::LmState) * kSlots + sizeof(unsigned long long) * kSlots;
        if (t_arena) slots.p = static_cast<vg::LmState *>(t_arena->pin_alloc(slots_bytes));
        if (!slots.p) {
            VG_HIP(hipHostMalloc(reinterpret_cast<void **>(&slots.p), slots_bytes, hipHostMallocCoherent));
            slots.owned = true;
        }
        slots.seq = reinterpret_cast<volatile unsigned long long *>(slots.p + kSlots);
        for (int k = 0; k < kSlots; k++) slots.seq[k] = 0ull;
        spin_wait = !vgi::debug_hook(vgi::kHookSolverEventWait);
        if (!spin_wait)
            for (auto &e : slots.ev) VG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return VG_OK;
    }

    // queue one LM iteration for parity `par` (current point = set / buffer `par`, candidate = the other one): pose rows + their
    // Gram, fixed-order sum (+ the Schur complement's all-reduce), reduced solve + back-substitution (+ the candidate's frames),
    // the candidate's evaluation, the accept kernel
    int queue_iteration(int par, bool gated, int &slot)
    {
        int rc;
        hipStream_t st = s.st;
        const int G = s.G, C = s.C;
        const int *gate = gated ? &d_state.p->gate : nullptr;
        vg::SchurArgs sa = s.schur_args(dset[par]);
        sa.mu = 0.;
        sa.mu_dev = &d_state.p->mu;
        sa.gate = gate;
        sa.gate_expect = par;
        if (s.n_poses) {
            // rows of every pose + the Gram of the rows, one launch; then ONE fixed-order sum over the workgroups
            hipLaunchKernelGGL(vg::vg_schur_rows_gram_kernel, dim3(s.sg_wgs), dim3(vg::kSchurThreads * s.sg_batches), s.sg_lds, st, sa, s.sg_ppw, s.sg_batches, s.d_rgroups.p, s.sg_shared);
            VG_HIP(hipGetLastError());
            vg::launch_strided_sum(st, s.d_rgroups.p, s.sg_wgs, C * C + 1, s.d_rgram.p);  // the Gram and the count of bad pose blocks
            VG_HIP(hipGetLastError());
        } else if (s.multi_rank) {
            // a rank without poses still joins the sum: the buffer holds the cross-rank total of the previous iteration
            VG_HIP(hipMemsetAsync(s.d_rgram.p, 0, sizeof(double) * s.h_rgram.size(), st));
        }
        VG_TRY(vgc::allreduce_sum(s.comm, s.d_rgram.p, s.h_rgram.size(), st));  // Schur complement of the poses of all ranks
        vg::LmSolveArgs r2 = ra;
        r2.gate_expect = gated ? par : -1;
        // every back-substitution workgroup solves the reduced system itself -- while there are few enough of them: the
        // redundant solves are SIMD time (~1 500 instructions per wave and workgroup), at 100 k poses (3 125 workgroups) they
        // made the launch 82 us where a one-workgroup solve launch + the plain back-substitution take 30
        const long long fold_max_groups = vgi::debug_hook(vgi::kHookSolverFoldMaxGroups) ? vgi::debug_hook(vgi::kHookSolverFoldMaxGroups) : vg::kFoldMaxGroups;
        const bool fold_solve = G > 0 && G <= vg::kFoldMaxG && (long long)s.n_bs_groups <= fold_max_groups;
        if (!fold_solve) {
            if (G <= vg::kEntrySolveMaxG)
                hipLaunchKernelGGL(vg::vg_lm_reduced_solve_entries_kernel, dim3(1), dim3(vg::kEntryThreads), sizeof(double) * vg::lm_entry_solve_lds_doubles(G), st, r2);
            else
                hipLaunchKernelGGL(vg::vg_lm_reduced_solve_kernel, dim3(1), dim3(G <= 64 ? vg::kWave : vg::kLmThreads), solve_lds, st, r2);
            VG_HIP(hipGetLastError());
        }
        vg::BacksubArgs ba;
        ba.s = sa;
        ba.dg = s.d_dg.p;
        ba.pose_param = s.d_pose_param.p;
        ba.gcol_param = s.d_gcol_param.p;
        ba.delta = s.d_delta.p;
        ba.scal = s.d_scal.p;
        ba.gmax_bits = s.d_gmax;
        ba.x = xbuf[par];
        ba.xg = s.d_xg;
        ba.lo = s.d_glo.p;
        ba.hi = s.d_ghi.p;
        ba.x_new = xbuf[1 - par];   // the step is applied where it is computed: no separate launch
        ba.fold = s.fold_frames ? s.d_fold.p : nullptr;   // ... and so are the candidate's frames
        ba.fold_gcol = s.d_fold_gcol.p;
        if (s.n_poses || G) {
            const unsigned int bs_grid = s.n_bs_groups ? s.n_bs_groups : 1u;
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
        s.p->gram_gate = gate;
        s.p->gram_gate_expect = par;
        // several ranks: the step's scalar sums are part of the evaluation's packed all-reduce (one rank: the accept kernel sums them)
        const int re = s.enqueue_evaluate(xbuf[1 - par], gset[1 - par], s.fold_frames && s.n_poses > 0, s.n_bs_groups && s.multi_rank);
        s.p->gram_gate = nullptr;
        if (re != VG_OK) return re;
        vg::LmAcceptArgs a2 = aa;
        a2.gate_expect = gated ? par : -1;
        slot = next_slot();
        arm_slot(slot, a2);
        hipLaunchKernelGGL(vg::vg_lm_accept_kernel, dim3(1), dim3(vg::kLmThreads), accept_lds, st, a2);
        VG_HIP(hipGetLastError());
        return queue_state(slot);
    }

    // the iterations: queue ahead, wait for the state of the iteration in flight, decide what the queue holds next
    int iterate(int &iter, int &parity, int &pending)
    {
        int rc;
        const vg_solve_options &opt = s.opt;
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
            const vg::LmState &S = slots.p[pending];
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
        return VG_OK;
    }

    int run()
    {
        int rc;
        hipStream_t st = s.st;
        const vg_solve_options &opt = s.opt;
        VG_TRY(prepare());
        if (t_arena) VG_TRY(t_arena->flush(st));  // every table of the set-up in one asynchronous copy
        s.mark("device-loop state");
        const double t_loop = now_s();  // everything before: allocation and upload of the problem's solver state
        VG_TRY(s.launch_init());   // clears, starting point into both parameter buffers, initial state
        VG_TRY(s.enqueue_evaluate(xbuf[0], gset[0]));
        int parity = 0, pending = next_slot(), iter = 0;
        arm_slot(pending, aa);
        hipLaunchKernelGGL(vg::vg_lm_accept_kernel, dim3(1), dim3(vg::kLmThreads), accept_lds, st, aa);
        VG_HIP(hipGetLastError());
        aa.init = 0;
        if (opt.max_num_iterations >= 1) VG_TRY(queue_iteration(parity, speculate, pending));
        else VG_TRY(queue_state(pending));
        VG_TRY(iterate(iter, parity, pending));
        VG_TRY(wait_state(pending));
        s.x_cur = xbuf[parity];       // the DevBuf handles keep their own buffers; the solve's current point is x_cur
        s.x_cand = xbuf[1 - parity];
        const vg::LmState S = slots.p[pending];
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
        return s.finish(iter, S.n_success, term, 0.5 * S.cost2_init, 0.5 * S.cost2, S.grad_max, S.radius, msg, true, t_loop);
    }
};

int LmSolve::run_device_loop()
{
    DeviceLoop loop(*this);
    return loop.run();
}

}  // namespace
