// vg_solver_device.hpp -- the two small kernels that keep a Levenberg-Marquardt iteration on the device
// (vg_solver_impl.hpp, "device loop"): the reduced (Schur) system with the active set of the box bounds, and the step
// acceptance / trust-region / convergence logic of Ceres' minimizer (what ceres::Solve does between two evaluations,
// src/calibration/unified_calibration.cpp:42-53 for the options).  With them an iteration is a fixed sequence of
// launches (+ two in-place RCCL all-reduces when sharded) and ONE host synchronisation -- the host only learns whether
// the step was accepted (it swaps the current / candidate buffers) and whether the solve is over.
// Both kernels are one workgroup: G <= 127 global columns.  Same arithmetic, in the same order, as the host versions
// they replace (chol_solve and the loop body of vg_problem_solve), so both loops walk the same iterates.
#pragma once

#include "vg_solver.hpp"

namespace vg {

struct LmState {
    double radius, decrease_factor, mu;
    double cost2, cost2_c;            // twice the cost at the current point / at the last candidate
    double grad_max, step_norm, rho, cost_change, model_change;
    int ucur;                          // which of the two U / g slots belongs to the current point
    int step_ok, accepted, done, term, iter, n_success, n_bad;
    int gate;                          // ucur while the solve runs, -1 once done: what speculatively queued launches test
    double cost2_init;                 // twice the cost at the starting point
};

struct LmSolveArgs {
    LmState *st;
    const double *U;      // [2][G*G]
    const double *gg;     // [2][G]
    const double *rgram;  // [(G+1)*(G+1)] Gram of the pose rows [L^-1 W^T | L^-1 g], summed over poses (and ranks)
    const double *lo, *hi;             // [G] box of every global column
    const unsigned char *gfrozen;      // [G] constant parameter blocks
    const double *xcur;                // [G] values of the global columns at the current point
    double *dg;                        // [G] out: global step
    double *S;                         // [G*G] global scratch for the damped reduced matrix, or NULL: it fits the LDS too
    int G, use_bounds;
    int gate_expect;                   // run only if st->gate == gate_expect (-1: no test)
    double dmin, dmax;
};

constexpr int kLmThreads = 256;

// dynamic LDS: A [G*G] | rhs [G] | b [G] | x [G] | held [G] (as double) ... + 2 flags
// Launched with ONE wave (64 threads) up to 64 columns -- the factorisation is a chain of G dependent column steps,
// and a workgroup barrier per step costs more than the step: 64 us at G = 45 with 256 threads, most of it barriers --
// and with 256 threads above.
__device__ __forceinline__ void lm_sync(int n_threads)
{
    if (n_threads == kWave) {  // one wave: its DS operations execute in order, only the compiler needs fencing
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// The solve itself, by the first kT threads of a workgroup (kT = 64: one wave, wave-level fences; else the whole workgroup);
// the step ends in LDS at sm + G * G + 2 * G (`x`); `publish`: also to a.dg / st->step_ok.
__device__ __forceinline__ void lm_reduced_solve_body(const LmSolveArgs &a, double *sm, const int tid, const int kT, const bool publish)
{
    const int G = a.G, C = G + 1;
    LmState *st = a.st;
    double *A = sm, *rhs = A + (size_t)G * G, *b = rhs + G, *x = b + G, *heldf = x + G, *flags = heldf + G;
    double *S = a.S ? a.S : flags + 2;  // the damped matrix survives the active-set passes: LDS when both fit
    const double mu = st->mu;
    const double *U = a.U + (size_t)st->ucur * G * G, *gg = a.gg + (size_t)st->ucur * G;
    for (int i = tid; i < G * G; i += kT) {
        const int r = i / G, c = i - r * G;
        double v = U[i] - a.rgram[(size_t)r * C + c];
        if (r == c) v += mu * clampd(U[i], a.dmin, a.dmax);
        S[i] = v;
    }
    for (int r = tid; r < G; r += kT) {
        rhs[r] = -gg[r] + a.rgram[(size_t)r * C + G];
        heldf[r] = a.gfrozen[r] ? 1. : 0.;
    }
    if (tid == 0) flags[0] = 1.;  // step_ok
    lm_sync(kT);
    // Constant blocks, and the active set of the box bounds: a parameter ON a bound whose step points outwards is held
    // (row and column leave the reduced system); re-solved until the set is stable.
    for (int pass = 0; pass <= G; pass++) {
        for (int i = tid; i < G * G; i += kT) {
            const int r = i / G, c = i - r * G;
            const bool h = heldf[r] != 0. || heldf[c] != 0.;
            A[i] = h ? (r == c ? 1. : 0.) : S[i];
        }
        for (int r = tid; r < G; r += kT) b[r] = heldf[r] != 0. ? 0. : rhs[r];
        lm_sync(kT);
        // Cholesky, lower triangle in place (right-looking; every entry sees its subtractions in increasing column
        // order, as the host's row-oriented version does)
        for (int j = 0; j < G; j++) {
            if (tid == 0) {
                const double d = A[(size_t)j * G + j];
                if (!(d > 0.) || !isfinite(d)) flags[0] = 0.;
                A[(size_t)j * G + j] = sqrt(d > 0. ? d : 1.);
            }
            lm_sync(kT);
            const double djj = A[(size_t)j * G + j];
            for (int i = j + 1 + tid; i < G; i += kT) A[(size_t)i * G + j] /= djj;
            lm_sync(kT);
            for (int i = j + 1 + tid; i < G; i += kT) {  // a row per lane: no index arithmetic, column j is a broadcast read
                const double lij = A[(size_t)i * G + j];
                for (int k = j + 1; k <= i; k++) A[(size_t)i * G + k] -= lij * A[(size_t)k * G + j];
            }
            lm_sync(kT);
        }
        // L y = b, L^T x = y
        for (int j = 0; j < G; j++) {
            if (tid == 0) b[j] = b[j] / A[(size_t)j * G + j];
            lm_sync(kT);
            const double yj = b[j];
            for (int i = j + 1 + tid; i < G; i += kT) b[i] -= A[(size_t)i * G + j] * yj;
            lm_sync(kT);
        }
        for (int j = G - 1; j >= 0; j--) {
            if (tid == 0) x[j] = b[j] / A[(size_t)j * G + j];
            lm_sync(kT);
            const double xj = x[j];
            for (int i = tid; i < j; i += kT) b[i] -= A[(size_t)j * G + i] * xj;
            lm_sync(kT);
        }
        if (tid == 0) flags[1] = 0.;  // changed
        lm_sync(kT);
        if (flags[0] != 0. && a.use_bounds)
            for (int r = tid; r < G; r += kT) {
                if (heldf[r] != 0.) continue;
                if ((a.xcur[r] <= a.lo[r] && x[r] < 0.) || (a.xcur[r] >= a.hi[r] && x[r] > 0.)) {
                    heldf[r] = 1.;
                    flags[1] = 1.;
                }
            }
        lm_sync(kT);
        if (flags[1] == 0. || flags[0] == 0.) break;
        lm_sync(kT);
    }
    if (publish) {
        for (int r = tid; r < G; r += kT) a.dg[r] = x[r];
        if (tid == 0) st->step_ok = (G == 0 || flags[0] != 0.) ? 1 : 0;
    }
}

#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(kLmThreads) void vg_lm_reduced_solve_kernel(LmSolveArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    LmState *st = a.st;
    if (st->done || (a.gate_expect >= 0 && st->gate != a.gate_expect)) return;
    lm_reduced_solve_body(a, sm, threadIdx.x, blockDim.x, true);
}
#endif

// Wide reduced systems (a rig: G = 45) in one workgroup of 256 threads, ENTRY-parallel: thread t owns the entries
// e = t + 256 m of the lower triangle of the augmented matrix [[S, .], [rhs^T, .]] (row G of its factor is y = L^-1 rhs: the
// forward substitution comes with the factorisation).  A column step is: every thread takes sqrt(A_jj) itself, the first G + 1
// threads scale column j, barrier, every owned entry (i, k > j) subtracts L_ij L_kj (independent LDS read-modify-writes),
// barrier.  The row-per-lane version above walks up to G entries of its row one after the other per column step: 107 us at
// G = 45 (rocprofv3), which is what kept the rig on the host-driven loop.  Each entry still sees its subtractions in
// increasing column order, square roots and divisions are IEEE: same bits as lm_reduced_solve_body / the host's chol_solve.
constexpr int kEntrySolveMaxG = 63;
constexpr int kEntrySolveSlots = ((kEntrySolveMaxG + 1) * (kEntrySolveMaxG + 2) / 2 + kLmThreads - 1) / kLmThreads;   // 9

__device__ __forceinline__ double readlane_f64(double v, int lane /* uniform */)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// dynamic LDS in doubles: A [C*C] | S [E] | Ld [G] | x [G] | held [G] | flag [2]
__host__ __device__ inline size_t lm_entry_solve_lds_doubles(int G)
{
    const size_t C = (size_t)G + 1;
    return C * C + C * (C + 1) / 2 + 3 * (size_t)G + 2;
}

#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(kLmThreads) void vg_lm_reduced_solve_entries_kernel(LmSolveArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    LmState *st = a.st;
    if (st->done || (a.gate_expect >= 0 && st->gate != a.gate_expect)) return;
    const int G = a.G, C = G + 1, E = C * (C + 1) / 2, tid = threadIdx.x;
    double *A = sm, *S = A + C * C, *Ld = S + E, *x = Ld + G, *heldf = x + G, *flag = heldf + G;
    const double mu = st->mu;
    const double *U = a.U + (size_t)st->ucur * G * G, *gg = a.gg + (size_t)st->ucur * G;
    int ri[kEntrySolveSlots], ck[kEntrySolveSlots];
#pragma unroll
    for (int m = 0; m < kEntrySolveSlots; m++) {
        const int e = tid + kLmThreads * m;
        int i = (int)((sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);
        while (i * (i + 1) / 2 > e) i--;
        while ((i + 1) * (i + 2) / 2 <= e) i++;
        const int k = e - i * (i + 1) / 2;
        const bool valid = e < E - 1;   // (G, G) is not used
        ri[m] = valid ? i : -1;
        ck[m] = valid ? k : 0;
        if (valid) {
            double v;
            if (i < G) {
                v = U[(size_t)i * G + k] - a.rgram[(size_t)i * C + k];
                if (i == k) v += mu * clampd(U[(size_t)i * G + i], a.dmin, a.dmax);
            } else {
                v = -gg[k] + a.rgram[(size_t)k * C + G];
            }
            S[e] = v;
        }
    }
    if (tid < G) heldf[tid] = a.gfrozen[tid] ? 1. : 0.;
    __syncthreads();
    bool ok = true;
    for (int pass = 0; pass <= G; pass++) {
        // constant blocks and the active set of the box bounds leave the system (unit row / column, zero right-hand side)
#pragma unroll
        for (int m = 0; m < kEntrySolveSlots; m++) {
            if (ri[m] < 0) continue;
            const int i = ri[m], k = ck[m];
            const bool h = (i < G && heldf[i] != 0.) || heldf[k] != 0.;
            A[i * C + k] = h ? ((i == k) ? 1. : 0.) : S[tid + kLmThreads * m];
        }
        if (tid == 0) flag[0] = 0.;
        __syncthreads();
        ok = true;
        for (int j = 0; j < G; j++) {
            const double d = A[j * C + j];   // never written during the factorisation: L_jj goes to Ld
            if (!(d > 0.) || !isfinite(d)) ok = false;
            const double sq = sqrt(d > 0. ? d : 1.);
            if (tid > j && tid <= G) A[tid * C + j] = A[tid * C + j] / sq;
            if (tid == j) Ld[j] = sq;
            __syncthreads();
#pragma unroll
            for (int m = 0; m < kEntrySolveSlots; m++) {
                if (ri[m] >= 0 && ck[m] > j) {
                    const int i = ri[m], k = ck[m];
                    A[i * C + k] -= A[i * C + j] * A[k * C + j];
                }
            }
            __syncthreads();
        }
        // L^T x = y (row G of the factor) from the last column up: first wave, the running right-hand side in registers
        if (tid < kWave) {
            double yy = tid < G ? A[G * C + tid] : 0., xv = 0.;
            const double dd = tid < G ? Ld[tid] : 1.;
            for (int j = G - 1; j >= 0; j--) {
                const double lj = tid < j ? A[j * C + tid] : 0.;
                const double xj = readlane_f64(yy, j) / readlane_f64(dd, j);
                yy -= lj * xj;
                if (tid == j) xv = xj;
            }
            if (tid < G) {
                x[tid] = xv;
                if (ok && a.use_bounds && heldf[tid] == 0. &&
                    ((a.xcur[tid] <= a.lo[tid] && xv < 0.) || (a.xcur[tid] >= a.hi[tid] && xv > 0.))) {
                    heldf[tid] = 1.;
                    flag[0] = 1.;
                }
            }
        }
        __syncthreads();
        if (flag[0] == 0. || !ok) break;
        __syncthreads();   // everybody has read the flag before the next pass clears it
    }
    if (tid < G) a.dg[tid] = x[tid];
    if (tid == 0) st->step_ok = (G == 0 || ok) ? 1 : 0;
}
#endif

// Narrow reduced systems (a mono or stereo calibration: G <= kFoldMaxG): EVERY workgroup of the back-substitution solves
// the G x G system itself (its first wave, in LDS, the arithmetic of vg_lm_reduced_solve_kernel: every workgroup gets the
// same bits) and goes on with the step in LDS -- one launch and one dependent round trip through HBM less per iteration; a
// 6 x 6 .. 24 x 24 Cholesky is a few microseconds of one wave next to the launch it replaces.  Workgroup 0 publishes the
// step and step_ok.
constexpr int kFoldMaxG = 24;

template <int kJ>
__global__ __launch_bounds__(kBsThreads) void vg_backsub_solve_kernel(BacksubArgs b, LmSolveArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    LmState *st = a.st;
    if (st->done || (a.gate_expect >= 0 && st->gate != a.gate_expect)) return;
    if (gate_closed(b.s.gate, b.s.gate_expect)) return;
    if (threadIdx.x < kWave) lm_reduced_solve_body(a, sm, threadIdx.x, kWave, blockIdx.x == 0);
    __syncthreads();
    backsub_body<kJ>(b, sm + (size_t)a.G * a.G + 2 * a.G);
}

struct LmAcceptArgs {
    LmState *st;
    double *U, *gg;                    // [2][G*G], [2][G]
    const double *sums;                // [n_ds][Wmax*Wmax] summed Gram blocks | [5] scalar sums of the step
    const int *inv;                    // [n_ds][G] global column -> local column of that dataset or -1
    const int *Wd;                     // [n_ds]
    const double *dg;                  // [G]
    unsigned long long *gmax_bits;     // max |g_pose| (bit pattern); reset here
    double *bad;                       // poses whose damped block was not positive definite, summed over ranks (it sits
                                       //   behind rgram and travelled with its all-reduce); reset here
    double *xcur;                      // [G] global values at the current point (updated on acceptance)
    const double *x;                   // init: the parameter vector, to gather xcur
    const long long *gcol_param;       // [G]
    const double *lo, *hi;
    const unsigned char *gfrozen;
    int n_ds, Wmax, G, init, multi_rank;
    const double *scal_partials;       // [n_scal][5] per-workgroup partials of the back-substitution's scalar sums, or NULL:
    unsigned int n_scal;               //   they were summed (and all-reduced) into sums[n_ds * Wmax^2 ..] by the caller
    int gate_expect;                   // run only if st->gate == gate_expect (-1: no test)
    // the host does not wait for an event behind this kernel (a marker packet in the stream: 5-6 us before the next iteration's
    // first kernel starts, rocprofv3 trace) but spins on host_seq: written with system-scope release AFTER the state
    unsigned long long *host_seq = nullptr;   // pinned; NULL: the host waits for an event (A/B hook)
    unsigned long long seq = 0;               // the value this launch leaves there
    LmState *host_state;               // pinned host memory (device-visible): the state after this call, for the host's
                                       //   decisions -- written by the kernel itself: a copy command between two kernels of
                                       //   a stream costs two engine hand-overs (~20 us), a store costs nothing
    size_t lds_doubles;                // dynamic LDS given to the launch, in doubles
    double dmin, dmax, ftol, gtol, ptol, min_rel_decrease, max_radius, min_radius;
};

#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(kLmThreads) void vg_lm_accept_kernel(LmAcceptArgs a)
{
    const int G = a.G, tid = threadIdx.x;
    LmState *st = a.st;
    if (st->done) {  // over before this launch (e.g. a non-finite cost at the starting point): the host still gets the state
        if (tid == 0 && a.host_state) {
            *a.host_state = *st;
            if (a.host_seq) __hip_atomic_store(a.host_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (a.gate_expect >= 0 && st->gate != a.gate_expect) return;
    const int slot = a.init ? st->ucur : 1 - st->ucur;  // where the freshly evaluated point's blocks go
    double *Uc = a.U + (size_t)slot * G * G, *gc = a.gg + (size_t)slot * G;
    const size_t WW = (size_t)a.Wmax * a.Wmax;
    // ONE round of global loads: everything below is requested before the first barrier -- the inputs of the scalar logic
    // (staged once: a dependent global load per loop step made this kernel 41 us at G = 45), the partials of the step's five
    // scalar sums, the state itself (thread 0) and the summed blocks.  (Four dependent rounds before: 7.4 us at G = 6.)
    __shared__ double s_ud[128], s_g[128], s_dg[128], s_x[128], s_lo[128], s_hi[128];
    __shared__ unsigned char s_fz[128];
    {
        const double *Ucur = a.U + (size_t)st->ucur * G * G, *gcur = a.gg + (size_t)st->ucur * G;
        for (int k = tid; k < G; k += kLmThreads) {
            s_ud[k] = a.init ? 0. : Ucur[(size_t)k * G + k];
            s_g[k] = a.init ? 0. : gcur[k];
            s_dg[k] = a.dg[k];
            s_x[k] = a.init ? 0. : a.xcur[k];
            s_lo[k] = a.lo[k];
            s_hi[k] = a.hi[k];
            s_fz[k] = a.gfrozen[k];
        }
    }
    // the five scalar sums of the step: summed here (fixed order) unless the caller already did (several ranks)
    __shared__ double s_sc[5], s_red[5][kLmThreads / kWave];
    if (a.scal_partials && !a.init) {
        double acc[5] = {0., 0., 0., 0., 0.};
        for (unsigned int i = tid; i < a.n_scal; i += kLmThreads)
#pragma unroll
            for (int q = 0; q < 5; q++) acc[q] += a.scal_partials[(size_t)i * 5 + q];
#pragma unroll
        for (int q = 0; q < 5; q++) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc[q] += __shfl_xor(acc[q], off, kWave);
            if ((tid & (kWave - 1)) == 0) s_red[q][tid >> 6] = acc[q];
        }
    }
    // The scalar logic runs on a LOCAL copy of the state: read once (together with the two counters), written back once.
    // Through the pointer every read after a write had to be a fresh global load (the compiler cannot rule out aliasing
    // with the counters): half a dozen dependent memory round trips in a one-thread section.
    LmState S;
    int n_bad = 0;                      // the GLOBAL count: every rank takes the same accept / reject branch
    unsigned long long gmax_bits = 0ull;
    if (tid == 0) {
        S = *st;
        n_bad = (int)*a.bad;
        gmax_bits = *a.gmax_bits;
    }
    // the column maps and the summed blocks are read many times: staged in LDS when they fit (a handful of datasets)
    extern __shared__ __attribute__((aligned(16))) double sm_acc[];
    const bool staged = a.lds_doubles >= (size_t)a.n_ds * WW + ((size_t)a.n_ds * G + 1) / 2 + 1;
    const double *sums = a.sums;
    const int *inv = a.inv;
    if (staged) {
        double *s_sums = sm_acc;
        int *s_inv = reinterpret_cast<int *>(sm_acc + (size_t)a.n_ds * WW);
        for (size_t i = tid; i < (size_t)a.n_ds * WW; i += kLmThreads) s_sums[i] = a.sums[i];
        for (int i = tid; i < a.n_ds * G; i += kLmThreads) s_inv[i] = a.inv[i];
        sums = s_sums;
        inv = s_inv;
    }
    __syncthreads();
    if (tid < 5 && a.scal_partials && !a.init) {
        double t = 0.;
        for (int w = 0; w < kLmThreads / kWave; w++) t += s_red[tid][w];
        s_sc[tid] = t;
    }
    // U, g of the evaluated point from the per-dataset sums, dataset after dataset (the order the host uses)
    for (int i = tid; i < G * G; i += kLmThreads) {
        const int ga = i / G, gb = i - ga * G;
        double s = 0.;
        for (int d = 0; d < a.n_ds; d++) {
            const int la = inv[d * G + ga], lb = inv[d * G + gb], W = a.Wd[d];
            if (la >= 0 && lb >= 0) s += sums[d * WW + (size_t)la * W + lb];
        }
        Uc[i] = s;
    }
    for (int ga = tid; ga < G; ga += kLmThreads) {
        double s = 0.;
        for (int d = 0; d < a.n_ds; d++) {
            const int la = inv[d * G + ga], W = a.Wd[d];
            if (la >= 0) s += sums[d * WW + (size_t)la * W + W - 1];
        }
        gc[ga] = s;
        if (a.init) a.xcur[ga] = a.x[a.gcol_param[ga]];
    }
    __syncthreads();
    if (tid != 0) return;
    double cost2_c = 0.;
    for (int d = 0; d < a.n_ds; d++) cost2_c += sums[d * WW + (size_t)a.Wd[d] * a.Wd[d] - 1];
    auto publish = [&]() {
        *st = S;
        if (a.host_state) {
            *a.host_state = S;
            if (a.host_seq) __hip_atomic_store(a.host_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    if (a.init) {
        S.cost2 = cost2_c;
        S.cost2_init = cost2_c;
        S.mu = 1. / S.radius;
        S.gate = S.ucur;
        // NaN / Inf in the residuals of the starting point: Ceres' evaluator fails the solve ("Residual and Jacobian
        // evaluation failed", TerminationType FAILURE) -- without this test the NaNs are silently dropped by the fmax of
        // the gradient norm and the solve "converges" at once
        if (!isfinite(cost2_c)) {
            S.done = 1;
            S.term = VG_TERM_FAILURE;
            S.gate = -1;
        }
        publish();
        return;
    }
    S.cost2_c = cost2_c;
    const double *sc = a.scal_partials ? s_sc : a.sums + (size_t)a.n_ds * WW;
    *a.bad = 0.;
    double gmax_p = __longlong_as_double((long long)gmax_bits);
    *a.gmax_bits = 0ull;
    S.n_bad += n_bad;
    const bool step_ok = S.step_ok != 0 && n_bad == 0;
    const double mu = S.mu;
    double rho = 0., step2 = 0., cost_change = 0., model_change = 0.;
    S.iter++;
    if (step_ok) {
        double xg2 = 0.;
        for (int k = 0; k < G; k++) xg2 += s_x[k] * s_x[k];
        const double gdp = sc[0], ddp = sc[1], dp2 = sc[2], gp2 = sc[3], xp2 = sc[4];
        // several ranks take the same branches only on summable quantities: the pose part of the gradient max-norm is
        // replaced by its 2-norm (an upper bound: the gradient test can only fire later than Ceres', never earlier)
        if (a.multi_rank) gmax_p = sqrt(gp2);
        double gdg = 0., ddg = 0., dg2 = 0., gmax_g = 0.;
        for (int k = 0; k < G; k++) {
            if (s_fz[k]) continue;
            const double dcl = clampd(s_ud[k], a.dmin, a.dmax);
            gdg += s_g[k] * s_dg[k];
            ddg += dcl * s_dg[k] * s_dg[k];
            dg2 += s_dg[k] * s_dg[k];
            // projected gradient for bounded parameters: |Project(x - g) - x|
            const double xv = s_x[k];
            const double xp = clampd(xv - s_g[k], s_lo[k], s_hi[k]);
            gmax_g = fmax(gmax_g, fabs(xp - xv));
        }
        S.grad_max = fmax(gmax_g, gmax_p);
        model_change = 0.5 * (mu * (ddg + ddp) - (gdg + gdp));  // 1/2 delta^T (mu D delta - g)
        step2 = dg2 + dp2;
        cost_change = 0.5 * (S.cost2 - cost2_c);
        rho = model_change > 0. ? cost_change / model_change : -1.;
        if (S.grad_max <= a.gtol) {
            S.term = VG_TERM_CONVERGENCE_GRADIENT;
            S.done = 1;
        } else if (sqrt(step2) <= a.ptol * (sqrt(xg2 + xp2) + a.ptol)) {
            S.term = VG_TERM_CONVERGENCE_PARAMETER;
            S.done = 1;
        }
    }
    S.rho = rho;
    S.step_norm = sqrt(step2);
    S.cost_change = cost_change;
    S.model_change = model_change;
    S.accepted = 0;
    if (S.done) {
        S.gate = -1;
        publish();
        return;
    }
    const bool success = step_ok && isfinite(cost2_c) && rho > a.min_rel_decrease;
    if (success) {
        S.n_success++;
        S.accepted = 1;
        S.ucur = 1 - S.ucur;
        for (int k = 0; k < G; k++) a.xcur[k] = clampd(s_x[k] + s_dg[k], s_lo[k], s_hi[k]);  // what the step kernel wrote
        const double prev = S.cost2;
        S.cost2 = cost2_c;
        const double t = 2. * rho - 1.;
        const double f = 1. - t * t * t;
        S.radius = fmin(S.radius / fmax(f, 1. / 3.), a.max_radius);
        S.decrease_factor = 2.;
        if (fabs(prev - cost2_c) <= a.ftol * prev) {
            S.term = VG_TERM_CONVERGENCE_FUNCTION;
            S.done = 1;
        }
    } else {
        S.radius /= S.decrease_factor;
        S.decrease_factor *= 2.;
        if (S.radius < a.min_radius) {
            S.term = VG_TERM_RADIUS_TOO_SMALL;
            S.done = 1;
        }
    }
    S.mu = 1. / S.radius;
    S.gate = S.done ? -1 : S.ucur;
    publish();
}
#endif

}  // namespace vg
