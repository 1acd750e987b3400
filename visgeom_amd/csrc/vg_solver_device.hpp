// vg_solver_device.hpp -- the two small kernels that keep a Levenberg-Marquardt iteration on the device
// (vg_solver_impl.hpp, "device loop"): the reduced (Schur) system with the active set of the box bounds, and the step
// acceptance / trust-region / convergence logic of Ceres' minimizer (what ceres::Solve does between two evaluations,
// src/calibration/unified_calibration.cpp:42-53 for the options).  With them an iteration is a fixed sequence of
// launches (+ two in-place RCCL all-reduces when sharded) and ONE host synchronisation -- the host only learns whether
// the step was accepted (it swaps the current / candidate buffers) and whether the solve is over.
// Both kernels are one workgroup: G <= 127 global columns.  The acceptance logic is the host loop's; the reduced solve up to 63
// columns is an L D L^T without square roots (lm_entry_solve_body) where the host runs a Cholesky: the two loops agree to
// rounding, not bit for bit -- with the reference's tolerances of 1e-15 (unified_calibration.cpp:47-49, below the noise of a
// cost summed over 10^6 residuals) the number of iterations at the tail of a solve differs between them, the optimum does not.
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

// Everything a solve clears or copies before its first evaluation, in ONE launch: four memset commands, two device-to-device
// copies and a pageable host-to-device copy of the initial state used to be seven blit operations of ~4.3 us each in front of the
// first kernel of every solve (rocprofv3: `fillBufferAligned` / `copyBuffer`, 40 us of a 0.52 ms solve).
struct SolverInitArgs {
    static constexpr int kZero = 6;
    double *zero[kZero] = {};            // ranges to clear (in doubles)
    unsigned long long n_zero[kZero] = {};
    const double *src = nullptr;         // the starting point ...
    double *dst0 = nullptr, *dst1 = nullptr;   // ... copied to the current and (device loop) the candidate buffer
    unsigned long long n_copy = 0;
    LmState *state = nullptr;            // device loop: the initial trust-region state, by value
    LmState h0;
    void add_zero(double *p, size_t n)
    {
        for (int k = 0; k < kZero; k++)
            if (!zero[k]) {
                zero[k] = p;
                n_zero[k] = n;
                return;
            }
    }
};

#ifdef VG_TU_SOLVER
__global__ __launch_bounds__(256) void vg_solver_init_kernel(SolverInitArgs a)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, step = (unsigned long long)gridDim.x * blockDim.x;
#pragma unroll
    for (int k = 0; k < SolverInitArgs::kZero; k++)
        for (unsigned long long i = t; i < a.n_zero[k]; i += step) a.zero[k][i] = 0.;
    for (unsigned long long i = t; i < a.n_copy; i += step) {
        const double v = a.src[i];
        a.dst0[i] = v;
        if (a.dst1) a.dst1[i] = v;
    }
    if (t == 0 && a.state) *a.state = a.h0;
}
#endif

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
    int one_wave = 0;                  // vg_backsub_solve_kernel: the first wave alone, a row per lane (the route before round 4; A/B hook)
};

constexpr int kLmThreads = 256;
constexpr int kLmMaxDatasets = kLmThreads;   // the accept kernel keeps the datasets' block widths in LDS, a thread each (more datasets: host-driven loop)

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

// Wide reduced systems (a rig: G = 45) in one workgroup of 1024 threads, ENTRY-parallel: a thread owns one to three entries of
// the lower triangle of the augmented matrix [[S, .], [rhs^T, .]] (row G ends as z = L~^-1 rhs: the forward substitution comes
// with the factorisation).  The factorisation is S = L~ D L~^T WITHOUT the scaling of a Cholesky column, TWO columns per step:
// with the pivots d0 = A_jj, b = A_j+1,j, d1 = A_j+1,j+1 - b^2 / d0 (every thread takes the reciprocals itself: v_rcp_f64 + two
// Newton steps) an owned entry (i, k > j + 1) loses A_ij A_kj / d0 + a_i a_k / d1, a_i = A_i,j+1 - A_ij b / d0 being the
// once-updated column j + 1 that every thread forms for its own rows; one barrier per step, no square root, no division.
// Then x_j = (z_j - sum_{i > j} A_ij x_i) / d_j from the last column up in the first wave (v_readlane broadcasts,
// reciprocals taken in parallel beforehand).
// History (G = 45, one active-set pass; rocprofv3 in the rig's loop / tools/exp/solve_kernel_probe.hip): a row per lane
// (lm_reduced_solve_body) 107 us; entry-parallel Cholesky with scaled columns, 256 threads (sqrt + division + two barriers per
// column) 40.9 us; L D L^T, one barrier per column, LDS reads of a step batched 34 us; 1024 threads 24 us; two columns per
// step 22 us.  What is left is 22 steps of ~1 500 cycles: a 16-wave barrier (~400), the LDS round trip in front of it and a
// chain of dependent FP64 operations at ~25 cycles each -- finished waves skipping the step and reciprocals side by side moved
// it by 3 % (profiles/NOTES.md).  Not the bits of the host's chol_solve (products associated differently): the two loops agree
// to rounding, which is what tests/test_gpu_solve.py::test_wide_system_on_the_device_resident_loop holds them to.
constexpr int kEntrySolveMaxG = 63;
constexpr int kEntryThreads = 1024;   // 16 waves, two entries per thread at G = 45: with 256 threads (five entries each, nine slots in the code) a column
                                      // step was ~150 instructions of ONE wave per SIMD, 1 300 cycles (tools/exp/solve_kernel_probe.hip)
constexpr int kEntrySolveSlots = ((kEntrySolveMaxG + 1) * (kEntrySolveMaxG + 2) / 2 + kEntryThreads - 1) / kEntryThreads;   // 3

__device__ __forceinline__ double readlane_f64(double v, int lane /* uniform */)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 1 / d for a positive, finite, normal d: hardware estimate + two Newton steps (<= 1 ulp or so; the IEEE division's
// scale / fix-up steps only matter for operands this solve rejects anyway)
__device__ __forceinline__ double rcp_pos(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.), r, r);
    r = fma(fma(-d, r, 1.), r, r);   // (v_rcp_f64 alone is good to ~2^-26: one step leaves ~2^-50, the second makes it a rounding error)
    return r;
}

// row of entry e of the lower triangle numbered row by row (e = i (i + 1) / 2 + k, k <= i); -1 for e < 0
__device__ __forceinline__ int entry_row(int e)
{
    if (e < 0) return -1;
    int i = (int)((sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);
    while (i * (i + 1) / 2 > e) i--;
    while ((i + 1) * (i + 2) / 2 <= e) i++;
    return i;
}

// row stride of the augmented matrix in LDS: odd, so that a column (what every step reads) spreads over the banks
__host__ __device__ inline int lm_entry_solve_stride(int G) { return (G + 1) | 1; }

// dynamic LDS in doubles: A [C * stride] | F [C * stride] | S [E] | x [G] | held [G] | flag [2]
__host__ __device__ inline size_t lm_entry_solve_lds_doubles(int G)
{
    const size_t C = (size_t)G + 1;
    return 2 * C * (size_t)lm_entry_solve_stride(G) + C * (C + 1) / 2 + 2 * (size_t)G + 2;
}

// -DVG_SOLVE_STAMPS (tools/exp/solve_kernel_probe.hip only): thread 0 stores the shader clock at the phase boundaries through
// a.S, which this kernel does not use otherwise
#ifdef VG_SOLVE_STAMPS
#define VG_SOLVE_STAMP(i)                                                                  \
    do {                                                                                   \
        if (threadIdx.x == 0 && a.S) reinterpret_cast<long long *>(a.S)[i] = (long long)clock64(); \
    } while (0)
#else
#define VG_SOLVE_STAMP(i) \
    do {                  \
    } while (0)
#endif

// The solve itself, by ALL kT threads of a workgroup (kSlots entries per thread at most); returns where the step lies in LDS;
// `publish`: also to a.dg / st->step_ok.  Used by the stand-alone kernel below (1024 threads, up to 63 columns) and by
// vg_backsub_solve_kernel (every back-substitution workgroup solves the system itself: 256 threads, up to kFoldMaxG columns).
template <int kT, int kSlots>
__device__ __forceinline__ double *lm_entry_solve_body(const LmSolveArgs &a, double *sm, const bool publish)
{
    LmState *st = a.st;
    const int G = a.G, C = G + 1, E = C * (C + 1) / 2, tid = threadIdx.x, P = lm_entry_solve_stride(G);
    double *A = sm, *F = A + C * P, *S = F + C * P, *x = S + E, *heldf = x + G, *flag = heldf + G;
    const double mu = st->mu;
    const double *U = a.U + (size_t)st->ucur * G * G, *gg = a.gg + (size_t)st->ucur * G;
    // Thread t owns the entries e = E - 2 - (t + 1024 m), m = 0, 1, ..: the LAST rows belong to the first wave.  Step j leaves the
    // rows <= j alone, so the waves of the high thread numbers finish first -- a finished wave only joins the step's barrier
    // (the pivot arithmetic every wave repeats is a third of a step's instructions) -- and wave 0, which also runs the
    // substitution, sees every pivot.
    int ri[kSlots], ck[kSlots], last_row[kSlots];   // last_row: highest row of the slot, uniform
#pragma unroll
    for (int m = 0; m < kSlots; m++) last_row[m] = entry_row(E - 2 - kT * m);
    const int wave_last_row = __builtin_amdgcn_readfirstlane(entry_row(E - 2 - (tid & ~(kWave - 1))));   // highest row of this wave
    {
        // every global load of the kernel is requested before the first one is used (a slot after the other: a dependent L2 round
        // trip per slot); invalid slots load entry (0, 0)
        double u_v[kSlots], r_v[kSlots];
#pragma unroll
        for (int m = 0; m < kSlots; m++) {
            const int e = E - 2 - (tid + kT * m);   // (G, G), entry E - 1, is not used
            const bool valid = e >= 0;
            const int i = valid ? entry_row(e) : 0;
            const int k = e - i * (i + 1) / 2;
            ri[m] = valid ? i : -1;
            ck[m] = valid ? k : 0;
            const int il = valid ? i : 0, kl = valid ? k : 0;
            u_v[m] = il < G ? U[(size_t)il * G + kl] : gg[kl];
            r_v[m] = il < G ? a.rgram[(size_t)il * C + kl] : a.rgram[(size_t)kl * C + G];
        }
#pragma unroll
        for (int m = 0; m < kSlots; m++) {
            if (ri[m] < 0) continue;
            double v;
            if (ri[m] < G) {
                v = u_v[m] - r_v[m];
                if (ri[m] == ck[m]) v += mu * clampd(u_v[m], a.dmin, a.dmax);
            } else {
                v = -u_v[m] + r_v[m];
            }
            S[tid + kT * m] = v;
        }
    }
    // the box test of the active set, per column (first wave): everything it reads from global memory, once
    double bx = 0., blo = 0., bhi = 0.;
    if (tid < G) {
        heldf[tid] = a.gfrozen[tid] ? 1. : 0.;
        if (a.use_bounds) {
            bx = a.xcur[tid];
            blo = a.lo[tid];
            bhi = a.hi[tid];
        }
    }
    __syncthreads();
    VG_SOLVE_STAMP(1);
    bool ok = true;
    for (int pass = 0; pass <= G; pass++) {
        // constant blocks and the active set of the box bounds leave the system (unit row / column, zero right-hand side)
#pragma unroll
        for (int m = 0; m < kSlots; m++) {
            if (ri[m] < 0) continue;
            const int i = ri[m], k = ck[m];
            const bool h = (i < G && heldf[i] != 0.) || heldf[k] != 0.;
            A[i * P + k] = h ? ((i == k) ? 1. : 0.) : S[tid + kT * m];
        }
        if (tid == 0) flag[0] = 0.;
        __syncthreads();
        VG_SOLVE_STAMP(2 + 3 * (pass < 2 ? pass : 2));
        ok = true;
        // TWO columns per step (a barrier and two LDS round trips per step are most of its ~950 cycles, tools/exp/solve_kernel_probe.hip):
        // with the pivots d0 = A_jj, b = A_j+1,j, d1 = A_j+1,j+1 - b^2 / d0 every thread forms the once-updated column j + 1 of its rows
        // itself (a_i1 = A_i,j+1 - A_ij b / d0) and subtracts both columns' products.  The final column j + 1 goes to a second
        // array F (other threads still read its old values in A during the step): odd columns of the factor live in F.
        int j = 0;
        for (; j + 1 < G; j += 2) {
            if (wave_last_row <= j) {   // nothing left in this wave's rows
                __syncthreads();
                continue;
            }
            const double d0 = A[j * P + j], b = A[(j + 1) * P + j], d1raw = A[(j + 1) * P + j + 1];
            double ai0[kSlots], ak0[kSlots], oi1[kSlots], ok1[kSlots], aik[kSlots];
            bool act[kSlots];
#pragma unroll
            for (int m = 0; m < kSlots; m++) {
                act[m] = false;
                if (last_row[m] <= j) continue;   // uniform over the workgroup
                act[m] = ri[m] >= 0 && ck[m] > j;
                if (!__any(act[m])) continue;     // a wave whose entries are all done issues nothing
                const int i = ri[m] < 0 ? 0 : ri[m], k = ck[m];
                ai0[m] = A[i * P + j];
                ak0[m] = A[k * P + j];
                oi1[m] = A[i * P + j + 1];
                ok1[m] = A[k * P + j + 1];
                aik[m] = A[i * P + k];
            }
            // the two reciprocals side by side (1 / d1 = d0 / (d0 d1raw - b^2)): a dependent FP64 operation is ~25 cycles here, and
            // the step is a chain of them between two barriers (wave 0: ~500 cycles of arithmetic per step before, probe)
            const double det = fma(d0, d1raw, -(b * b));
            if (!(d0 > 0.) || !isfinite(d0) || !(det > 0.) || !isfinite(det)) ok = false;
            const double r0 = rcp_pos(d0 > 0. ? d0 : 1.);
            const double r1 = d0 * rcp_pos(det > 0. ? det : 1.);
            const double l = b * r0;
#pragma unroll
            for (int m = 0; m < kSlots; m++) {
                if (!act[m]) continue;
                const double ai1 = fma(-ai0[m], l, oi1[m]);
                if (ck[m] == j + 1) {
                    F[ri[m] * P + j + 1] = ai1;
                } else {
                    const double ak1 = fma(-ak0[m], l, ok1[m]);
                    A[ri[m] * P + ck[m]] = fma(-ai1, ak1 * r1, fma(-ai0[m], ak0[m] * r0, aik[m]));
                }
            }
            __syncthreads();
        }
        if (j < G) {   // an odd number of columns: the last one is alone, stays in A (an even column) and has nothing below it but z
            const double d = A[j * P + j];
            if (!(d > 0.) || !isfinite(d)) ok = false;
        }
        VG_SOLVE_STAMP(3 + 3 * (pass < 2 ? pass : 2));
        // D L~^T x = z (row G) from the last column up: first wave, the running right-hand side in registers; lane t holds
        // column t (odd columns: in F)
        if (tid < kWave) {
            const double *Lt = A + ((tid & 1) ? C * P : 0);
            double yy = tid < G ? Lt[G * P + tid] : 0., xv = 0.;
            const double dd = tid < G ? Lt[tid * P + tid] : 1.;
            const double rdd = 1. / (dd > 0. ? dd : 1.);
            double lj = (G > 0 && tid < G - 1) ? Lt[(G - 1) * P + tid] : 0.;
            for (int q = G - 1; q >= 0; q--) {
                const double l_next = (q > 0 && tid < q - 1) ? Lt[(q - 1) * P + tid] : 0.;   // the next step's row: off the dependent chain
                const double xq = readlane_f64(yy, q) * readlane_f64(rdd, q);
                yy -= lj * xq;
                if (tid == q) xv = xq;
                lj = l_next;
            }
            if (tid < G) {
                x[tid] = xv;
                if (ok && a.use_bounds && heldf[tid] == 0. && ((bx <= blo && xv < 0.) || (bx >= bhi && xv > 0.))) {
                    heldf[tid] = 1.;
                    flag[0] = 1.;
                }
            }
            if (tid == 0) flag[1] = ok ? 1. : 0.;   // the first wave saw every pivot
        }
        __syncthreads();
        VG_SOLVE_STAMP(4 + 3 * (pass < 2 ? pass : 2));
        if (flag[0] == 0. || flag[1] == 0.) break;
        __syncthreads();   // everybody has read the flag before the next pass clears it
    }
    if (publish) {
        if (tid < G) a.dg[tid] = x[tid];
        if (tid == 0) st->step_ok = (G == 0 || ok) ? 1 : 0;
    }
    VG_SOLVE_STAMP(11);
    return x;
}

#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(kEntryThreads) void vg_lm_reduced_solve_entries_kernel(LmSolveArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    LmState *st = a.st;
    VG_SOLVE_STAMP(0);
    if (st->done || (a.gate_expect >= 0 && st->gate != a.gate_expect)) return;
    lm_entry_solve_body<kEntryThreads, kEntrySolveSlots>(a, sm, true);
}
#endif

// Narrow reduced systems (a mono or stereo calibration: G <= kFoldMaxG): EVERY workgroup of the back-substitution solves
// the G x G system itself (all 256 threads, lm_entry_solve_body in LDS: every workgroup gets the same bits) and goes on with
// the step in LDS -- one launch and one dependent round trip through HBM less per iteration.  Workgroup 0 publishes the step
// and step_ok.  Until round 4 the first wave did it alone with a row per lane: 17 of this kernel's 33.7 us for the stereo problem
// (G = 18), where the 256-thread L D L^T needs ~3; same box, per LM iteration: stereo 0.111 -> 0.094 ms, Mei (G = 10) 0.101 -> 0.090.
constexpr int kFoldMaxG = 24;
constexpr int kFoldMaxGroups = 480;    // back-substitution workgroups (32 poses each) up to which each of them repeats the reduced solve
constexpr int kFoldSlots = ((kFoldMaxG + 1) * (kFoldMaxG + 2) / 2 + kBsThreads - 1) / kBsThreads;   // 2

template <int kJ, bool kFrames>
__global__ __launch_bounds__(kBsThreads) void vg_backsub_solve_kernel(BacksubArgs b, LmSolveArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    LmState *st = a.st;
    if (st->done || (a.gate_expect >= 0 && st->gate != a.gate_expect)) return;
    if (gate_closed(b.s.gate, b.s.gate_expect)) return;
    // (round 4: the entry-parallel L D L^T on all 256 threads; before, the first wave alone with a row per lane -- 17 of the
    //  stereo problem's 33.7 us in this kernel at G = 18)
    const double *step;
    if (a.one_wave) {
        if (threadIdx.x < kWave) lm_reduced_solve_body(a, sm, threadIdx.x, kWave, blockIdx.x == 0);
        step = sm + (size_t)a.G * a.G + 2 * a.G;
    } else {
        step = lm_entry_solve_body<kBsThreads, kFoldSlots>(a, sm, blockIdx.x == 0);
    }
    __syncthreads();
    backsub_body<kJ, kFrames>(b, step);
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
    const size_t WW = (size_t)a.Wmax * a.Wmax;
    // ONE round of global loads, requested before anything is tested: the state (every thread: one line), the inputs of the
    // scalar logic for BOTH slots of U / g (which one is current is in the state), the partials of the step's five scalar sums,
    // the counters, the column maps and the summed blocks.  History: a dependent global load per loop step 41 us at G = 45;
    // four dependent rounds 7.4 us at G = 6; state -> current slot -> blocks as three rounds 22.7 / 14 us at G = 45.
    static_assert(kLmThreads >= 128, "a thread per global column");
    __shared__ double s_ud[128], s_g[128], s_dg[128], s_x[128], s_lo[128], s_hi[128];
    __shared__ unsigned char s_fz[128];
    __shared__ int s_W[kLmMaxDatasets];   // the datasets' block widths: read in every step of the loops below (as a global load each: 16 of 22.7 us)
    __shared__ double s_sc[5], s_red[5][kLmThreads / kWave];
    extern __shared__ __attribute__((aligned(16))) double sm_acc[];
    const int st_done = st->done, st_gate = st->gate, st_ucur = st->ucur;
    double c_ud[2] = {0., 0.}, c_g[2] = {0., 0.}, c_dg = 0., c_x = 0., c_lo = 0., c_hi = 0.;
    unsigned char c_fz = 0;
    if (tid < G) {
        if (!a.init) {
            c_ud[0] = a.U[(size_t)tid * G + tid];
            c_ud[1] = a.U[(size_t)G * G + (size_t)tid * G + tid];
            c_g[0] = a.gg[tid];
            c_g[1] = a.gg[G + tid];
            c_x = a.xcur[tid];
        }
        c_dg = a.dg[tid];
        c_lo = a.lo[tid];
        c_hi = a.hi[tid];
        c_fz = a.gfrozen[tid];
    }
    const int c_W = tid < a.n_ds ? a.Wd[tid] : 0;
    // the five scalar sums of the step: summed here (fixed order) unless the caller already did (several ranks)
    double acc[5] = {0., 0., 0., 0., 0.};
    if (a.scal_partials && !a.init)
        for (unsigned int i = tid; i < a.n_scal; i += kLmThreads)
#pragma unroll
            for (int q = 0; q < 5; q++) acc[q] += a.scal_partials[(size_t)i * 5 + q];
    // The scalar logic runs on a LOCAL copy of the state: read once (together with the two counters), written back once.
    // Through the pointer every read after a write had to be a fresh global load (the compiler cannot rule out aliasing
    // with the counters): half a dozen dependent memory round trips in a one-thread section.
    // The FIRST WAVE runs it, every lane on the same values (the sums over the global columns and the update of their current
    // values are spread over its lanes; lane 0 alone stores the state).
    LmState S;
    int n_bad = 0;                      // the GLOBAL count: every rank takes the same accept / reject branch
    unsigned long long gmax_bits = 0ull;
    if (tid < kWave) {
        S = *st;
        n_bad = (int)*a.bad;
        gmax_bits = *a.gmax_bits;
    }
    // the column maps and the summed blocks are read many times: staged in LDS when they fit (a handful of datasets)
    const bool staged = a.lds_doubles >= (size_t)a.n_ds * WW + ((size_t)a.n_ds * G + 1) / 2 + 1;
    const size_t inv_off = (size_t)a.n_ds * WW;
    if (staged) {
        int *s_inv = reinterpret_cast<int *>(sm_acc + inv_off);
        // eight loads in flight per thread: as "load, store, next" every step of this loop waited a whole L2 round trip
        // (9 steps at G = 45: most of the kernel's 13 us)
        const size_t n_sums = (size_t)a.n_ds * WW;
        for (size_t base = 0; base < n_sums; base += (size_t)8 * kLmThreads) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const size_t i = base + (size_t)q * kLmThreads + tid;
                v[q] = i < n_sums ? a.sums[i] : 0.;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const size_t i = base + (size_t)q * kLmThreads + tid;
                if (i < n_sums) sm_acc[i] = v[q];
            }
        }
        for (int i = tid; i < a.n_ds * G; i += kLmThreads) s_inv[i] = a.inv[i];
    }
    if (st_done) {  // over before this launch (e.g. a non-finite cost at the starting point): the host still gets the state
        if (tid == 0 && a.host_state) {
            *a.host_state = S;
            if (a.host_seq) __hip_atomic_store(a.host_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (a.gate_expect >= 0 && st_gate != a.gate_expect) return;
    const int slot = a.init ? st_ucur : 1 - st_ucur;  // where the freshly evaluated point's blocks go
    double *Uc = a.U + (size_t)slot * G * G, *gc = a.gg + (size_t)slot * G;
    if (tid < G) {
        s_ud[tid] = c_ud[st_ucur & 1];
        s_g[tid] = c_g[st_ucur & 1];
        s_dg[tid] = c_dg;
        s_x[tid] = c_x;
        s_lo[tid] = c_lo;
        s_hi[tid] = c_hi;
        s_fz[tid] = c_fz;
    }
    if (tid < a.n_ds) s_W[tid] = c_W;
    if (a.scal_partials && !a.init) {
#pragma unroll
        for (int q = 0; q < 5; q++) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc[q] += __shfl_xor(acc[q], off, kWave);
            if ((tid & (kWave - 1)) == 0) s_red[q][tid >> 6] = acc[q];
        }
    }
    __syncthreads();
    if (tid < 5 && a.scal_partials && !a.init) {
        double t = 0.;
        for (int w = 0; w < kLmThreads / kWave; w++) t += s_red[tid][w];
        s_sc[tid] = t;
    }
    // U, g of the evaluated point from the per-dataset sums, dataset after dataset (the order the host uses).  Twice in the
    // code, once on the LDS copies and once on global memory: through ONE pair of pointers that may be either, every read is a
    // flat load that the stores of U have to be ordered against.
    double cost2_c = 0.;
    auto assemble = [&](auto sums_at, auto inv_at) {
        for (int i = tid; i < G * G; i += kLmThreads) {
            const int ga = i / G, gb = i - ga * G;
            double s = 0.;
            for (int d = 0; d < a.n_ds; d++) {
                const int la = inv_at(d * G + ga), lb = inv_at(d * G + gb), W = s_W[d];
                if (la >= 0 && lb >= 0) s += sums_at(d * WW + (size_t)la * W + lb);
            }
            Uc[i] = s;
        }
        for (int ga = tid; ga < G; ga += kLmThreads) {
            double s = 0.;
            for (int d = 0; d < a.n_ds; d++) {
                const int la = inv_at(d * G + ga), W = s_W[d];
                if (la >= 0) s += sums_at(d * WW + (size_t)la * W + W - 1);
            }
            gc[ga] = s;
        }
    };
    // (indexing the shared array itself: an LDS POINTER handed to the lambda becomes a flat pointer whose null test this
    // compiler cannot select -- "V_CMP_NE_U32 0, src_shared_base: operand has incorrect register class")
    if (staged) assemble([&](size_t i) { return sm_acc[i]; }, [&](int i) { return reinterpret_cast<const int *>(sm_acc + inv_off)[i]; });
    else assemble([&](size_t i) { return a.sums[i]; }, [&](int i) { return a.inv[i]; });
    // the blocks' last entries: the datasets' squared residual norms
    if (staged) {
        for (int d = 0; d < a.n_ds; d++) {
            const int W = s_W[d];
            cost2_c += sm_acc[d * WW + (size_t)W * W - 1];
        }
    } else {
        for (int d = 0; d < a.n_ds; d++) cost2_c += a.sums[d * WW + (size_t)s_W[d] * s_W[d] - 1];
    }
    if (a.init)
        for (int ga = tid; ga < G; ga += kLmThreads) a.xcur[ga] = a.x[a.gcol_param[ga]];
    __syncthreads();
    if (tid >= kWave) return;
    auto publish = [&]() {
        if (tid != 0) return;
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
    double gmax_p = __longlong_as_double((long long)gmax_bits);
    if (tid == 0) {
        *a.bad = 0.;
        *a.gmax_bits = 0ull;
    }
    S.n_bad += n_bad;
    const bool step_ok = S.step_ok != 0 && n_bad == 0;
    const double mu = S.mu;
    double rho = 0., step2 = 0., cost_change = 0., model_change = 0.;
    S.iter++;
    if (step_ok) {
        // sums over the global columns: a lane per column (two above 64), butterfly over the wave -- a fixed order, every lane
        // ends with the same bits
        double xg2 = 0., gdg = 0., ddg = 0., dg2 = 0., gmax_g = 0.;
        for (int k = tid; k < G; k += kWave) {
            xg2 += s_x[k] * s_x[k];
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
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            xg2 += __shfl_xor(xg2, off, kWave);
            gdg += __shfl_xor(gdg, off, kWave);
            ddg += __shfl_xor(ddg, off, kWave);
            dg2 += __shfl_xor(dg2, off, kWave);
            gmax_g = fmax(gmax_g, __shfl_xor(gmax_g, off, kWave));
        }
        const double gdp = sc[0], ddp = sc[1], dp2 = sc[2], gp2 = sc[3], xp2 = sc[4];
        // several ranks take the same branches only on summable quantities: the pose part of the gradient max-norm is
        // replaced by its 2-norm (an upper bound: the gradient test can only fire later than Ceres', never earlier)
        if (a.multi_rank) gmax_p = sqrt(gp2);
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
        } else if (model_change > 0. && isfinite(cost2_c) && fabs(S.cost2 - cost2_c) <= a.ftol * S.cost2) {
            // Ceres tests the function tolerance on every evaluated candidate of a valid step, BEFORE the step is accepted or
            // rejected, and returns at the current point (see the host loop)
            S.term = VG_TERM_CONVERGENCE_FUNCTION;
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
        for (int k = tid; k < G; k += kWave) a.xcur[k] = clampd(s_x[k] + s_dg[k], s_lo[k], s_hi[k]);  // what the step kernel wrote
        S.cost2 = cost2_c;
        const double t = 2. * rho - 1.;
        const double f = 1. - t * t * t;
        S.radius = fmin(S.radius / fmax(f, 1. / 3.), a.max_radius);
        S.decrease_factor = 2.;
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
