// vg_solver.hpp -- device kernels of the Levenberg-Marquardt driver (what replaces
// ceres::Solve at src/calibration/unified_calibration.cpp:53 for the calibration problem).
//
// The normal equations have arrow structure: a small dense global block (all intrinsics + global transforms,
// G columns) and one 6x6 block per pose instance of a sequence transform, coupled only through the global block:
//     [ U   W ] [dg]     [gg]          U = sum A^T A,  V_i = sum B^T B,  W_i = sum A^T B,   J_b = [A_b | B_b]
//     [ W^T V ] [dp] = - [gp]
// Every pose is eliminated on the GPU (one lane per pose): V_i' = V_i + mu*D_i = L L^T, and the six rows
//     z_k = (L^-1 W_i^T)[k, :],   y_k = (L^-1 gp_i)[k]
// are emitted.  The Schur complement is then a Gram sum over those rows,
//     S = U' - sum_rows z z^T,   rhs = -gg + sum_rows z y,
// contracted with the same fixed-order FP64-MFMA machinery as the per-image Grams.  The host factors the G x G
// system (G <= ~50), and a second lane-per-pose kernel back-substitutes dp_i = -V_i'^-1 (gp_i + W_i^T dg).
#pragma once

#include "vg_gram.hpp"

namespace vg {

constexpr int kMaxLocalCols = 10 + 6 * kMaxChain;
constexpr int kPoseRec = 34;  // per pose: L (21, packed lower) | gp (6) | diag V (6) | active (1)

struct SolveDatasetDev {
    const double *gram;  // [n_blocks][W*W] at the current point
    int W;
    int pose_off;        // local column of the sequence member, -1 when the chain has none
};

__device__ __forceinline__ int tri(int r, int c) { return r * (r + 1) / 2 + c; }

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// y = L^-1 w  (L packed lower 6x6)
__device__ __forceinline__ void fwd6(const double *L, const double *w, double *y)
{
#pragma unroll
    for (int r = 0; r < 6; r++) {
        double s = w[r];
#pragma unroll
        for (int c = 0; c < r; c++) s -= L[tri(r, c)] * y[c];
        y[r] = s / L[tri(r, r)];
    }
}

// x = L^-T y
__device__ __forceinline__ void bwd6(const double *L, const double *y, double *x)
{
#pragma unroll
    for (int r = 5; r >= 0; r--) {
        double s = y[r];
#pragma unroll
        for (int c = r + 1; c < 6; c++) s -= L[tri(c, r)] * x[c];
        x[r] = s / L[tri(r, r)];
    }
}

struct SchurArgs {
    const SolveDatasetDev *ds;
    const int *inv;        // [n_datasets][G]: global column -> local column of that dataset, or -1
    const int *ref_ptr;    // [n_poses + 1]
    const int *ref_ds;     // [n_refs]
    const int *ref_blk;    // [n_refs]
    const unsigned char *pose_frozen;
    int G;
    int n_poses;
    double mu, dmin, dmax;
    const double *mu_dev;  // device loop: the damping parameter lives in the LM state (overrides mu when not NULL)
    double *rec;           // [n_poses][kPoseRec]
    double *rows;          // [n_poses * 6][G + 1]
    double *bad;           // number of poses whose damped block was not positive definite: a double in the slot behind the
                           // Schur Gram (rgram[(G+1)^2]), so that it is summed over ranks by that buffer's all-reduce
    const int *gate;       // speculative launches of the device loop: run only if *gate == gate_expect (NULL: always)
    int gate_expect;
    int n_ds = 0;          // entries of `ds` (vg_schur_rows_gram_kernel keeps up to kSchurLdsDatasets of them in LDS; 0: reads them from memory)
    unsigned long long *zero_u64 = nullptr;  // vg_schur_rows_gram_kernel clears this word (the max |g_pose| of the step that
                                             // follows): no memset command between two kernels of the host-driven loop
};

// The elimination of pose i: V_i, g_i gathered from the Gram blocks of every dataset that references the pose, damping,
// 6 x 6 Cholesky.  Evaluated by EVERY lane of the pose in the rows kernel below (a 6 x 6 factorisation is ~150
// instructions; a separate one-lane-per-pose launch in front of the rows kernel cost 9 us of launch + latency per
// iteration for a few microseconds of arithmetic).
struct PoseFactor {
    double L[21], gp[6], vd[6];
    bool active, pd;
    unsigned char mode;
};

// damping + 6 x 6 Cholesky of a pose block whose V_i (packed lower, 21) and g_i (6) have been gathered; has_refs: the pose
// is referenced by at least one residual block
// (mode = a.pose_frozen[i] and mu are arguments so that a caller can request them with its first loads)
__device__ __forceinline__ void pose_factor_from(const SchurArgs &a, unsigned char mode, double mu, double (&V)[21], bool has_refs, PoseFactor &f)
{
    // pose_frozen: 0 = eliminated here, 1 = constant, 2 = belongs to a sequence coupled by OdometryPrior blocks:
    // its raw V_i / g_i go to the record and the host eliminates the whole sequence as a block-tridiagonal system
    f.mode = mode;
    f.pd = true;
    if (f.mode == 2) {
#pragma unroll
        for (int k = 0; k < 21; k++) f.L[k] = V[k];
#pragma unroll
        for (int k = 0; k < 6; k++) f.vd[k] = V[tri(k, k)];
        f.active = false;
        return;
    }
    f.active = f.mode == 0 && has_refs;
#pragma unroll
    for (int r = 0; r < 6; r++) {
        f.vd[r] = V[tri(r, r)];
        V[tri(r, r)] += mu * clampd(f.vd[r], a.dmin, a.dmax);
    }
    // in-place Cholesky, packed lower
#pragma unroll
    for (int r = 0; r < 6; r++) {
#pragma unroll
        for (int c = 0; c <= r; c++) {
            double s = V[tri(r, c)];
#pragma unroll
            for (int k = 0; k < c; k++) s -= f.L[tri(r, k)] * f.L[tri(c, k)];
            if (r == c) {
                if (!(s > 0.)) { f.pd = false; s = 1.; }
                f.L[tri(r, r)] = sqrt(s);
            } else {
                f.L[tri(r, c)] = s / f.L[tri(c, c)];
            }
        }
    }
    if (f.active && !f.pd) f.active = false;
}

__device__ __forceinline__ void pose_factor(const SchurArgs &a, int i, PoseFactor &f)
{
    const int r0 = a.ref_ptr[i], r1 = a.ref_ptr[i + 1];
    double V[21];
#pragma unroll
    for (int k = 0; k < 21; k++) V[k] = 0.;
#pragma unroll
    for (int k = 0; k < 6; k++) f.gp[k] = 0.;
    for (int q = r0; q < r1; q++) {
        const SolveDatasetDev D = a.ds[a.ref_ds[q]];
        const double *Gb = D.gram + (size_t)a.ref_blk[q] * D.W * D.W;
        const int o = D.pose_off;
#pragma unroll
        for (int r = 0; r < 6; r++) {
#pragma unroll
            for (int c = 0; c <= r; c++) V[tri(r, c)] += Gb[(o + r) * D.W + o + c];
            f.gp[r] += Gb[(o + r) * D.W + D.W - 1];
        }
    }
    pose_factor_from(a, a.pose_frozen[i], a.mu_dev ? *a.mu_dev : a.mu, V, r1 > r0, f);
}

// one lane per (pose, global column): column gcol of the six rows  L^-1 W_i^T  (gcol == G: L^-1 g_i; that lane also
// leaves the pose's record [L | g | diag V | active] for the back-substitution and counts non-positive-definite blocks)
#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_schur_rows_kernel(SchurArgs a)
{
    const int C = a.G + 1;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gate_closed(a.gate, a.gate_expect)) return;
    if (t >= (long long)a.n_poses * C) return;
    const int i = (int)(t / C), gcol = (int)(t - (long long)i * C);
    PoseFactor f;
    pose_factor(a, i, f);
    double w[6], y[6];
    if (gcol < a.G) {
#pragma unroll
        for (int c = 0; c < 6; c++) w[c] = 0.;
        const int r0 = a.ref_ptr[i], r1 = a.ref_ptr[i + 1];
        for (int q = r0; q < r1; q++) {
            const int d = a.ref_ds[q];
            const int lc = a.inv[d * a.G + gcol];
            if (lc < 0) continue;
            const SolveDatasetDev D = a.ds[d];
            const double *Gb = D.gram + (size_t)a.ref_blk[q] * D.W * D.W;
#pragma unroll
            for (int c = 0; c < 6; c++) w[c] += Gb[lc * D.W + D.pose_off + c];
        }
    } else {
#pragma unroll
        for (int c = 0; c < 6; c++) w[c] = f.gp[c];
        double *rec = a.rec + (size_t)i * kPoseRec;
#pragma unroll
        for (int k = 0; k < 21; k++) rec[k] = f.L[k];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            rec[21 + k] = f.gp[k];
            rec[27 + k] = f.vd[k];
        }
        rec[33] = f.active ? 1. : 0.;
        if (f.mode == 0 && !f.pd && a.ref_ptr[i + 1] > a.ref_ptr[i]) atomicAdd(a.bad, 1.);
    }
    double *out = a.rows + (size_t)i * 6 * C + gcol;
    if (f.mode == 2) {  // host-eliminated sequence: hand over the raw column of W_i^T (last column: g_i)
#pragma unroll
        for (int k = 0; k < 6; k++) out[k * C] = w[k];
        return;
    }
    fwd6(f.L, w, y);
#pragma unroll
    for (int k = 0; k < 6; k++) out[k * C] = f.active ? y[k] : 0.;
}
#endif

// vg_schur_rows_kernel AND the Gram of the rows it produces, in one launch: a workgroup owns `poses_per_wg` whole poses per
// batch (one lane per (pose, column), as above), keeps the batch's 6 x poses_per_wg rows in LDS next to writing them out, and
// adds their outer products into its own C x C partial (entry-parallel, rows in increasing order: fixed order; entry C * C
// of the partial = the workgroup's count of blocks that were not positive definite); `batches` batches per workgroup keep the
// number of partials in the hundreds.  One strided fixed-order sum over the C * C + 1 entries of the workgroup partials then
// gives the Schur complement's Gram and the count -- two launches where there were three (rows, MFMA Gram per 96 rows,
// sum), and the rows are read back from LDS instead of from HBM.  Not for host-eliminated sequences (mode 2): their rows are
// rewritten by the host before the Gram.
constexpr int kSchurThreads = 256;      // lanes of one batch: (pose of the batch, column)
constexpr int kSchurMaxBatches = 4;    // batches of a workgroup run side by side: blockDim.x = kSchurThreads * batches
constexpr int kSchurMaxRefs = 256;     // references of a workgroup's poses resolved into LDS (beyond: read per lane from memory)
constexpr int kSchurLdsDatasets = 32;  // dataset descriptors kept in LDS (requested with the kernel's first loads)

#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(kSchurThreads * kSchurMaxBatches) void vg_schur_rows_gram_kernel(SchurArgs a, int poses_per_wg, int batches,
                                                                            double *__restrict__ partials /* [n_wg][C*C + 1] */,
                                                                            int shared_gather)
{
    extern __shared__ __attribute__((aligned(16))) double sm_rows[];  // [batches * poses_per_wg * 6][C + 1] (odd-ish stride)
#ifdef VG_SCHUR_EVEN_STRIDE   // A/B library: the stride before round 6
    const int C = a.G + 1, CS = C + 1, tid = threadIdx.x % kSchurThreads;
#else
    const int C = a.G + 1, CS = C | 1, tid = threadIdx.x % kSchurThreads;   // odd row stride (the LDS is sized for C + 1): with C = 7 a stride of 8 doubles put every fourth row on the same banks
#endif
    const int pl = tid / C, gcol = tid - pl * C;        // pose of the batch, column
    const bool lane_on = pl < poses_per_wg;
    // every batch has its own 256 lanes (blockDim.x = kSchurThreads * batches): the chains pose -> reference list -> Gram
    // blocks -> factorisation of the batches overlap instead of following one another (four in a row were 13 us of a wave's
    // life, 73 % of it waiting)
    const int bt = threadIdx.x / kSchurThreads;
    const int i = (blockIdx.x * batches + bt) * poses_per_wg + pl;
    const bool pose_on = lane_on && bt < batches && i < a.n_poses;
    // ---- the kernel's FIRST round of loads, all of them requested before the gate is tested: the gate itself, the bounds of the
    // workgroup's and of this lane's reference lists, the pose's mode, the damping parameter, the dataset descriptors (to LDS).
    // Each used to be its own dependent round trip at the place it was needed (~1 us each with one workgroup per CU).
    const int i_first = blockIdx.x * batches * poses_per_wg;
    int i_end = i_first + batches * poses_per_wg;
    i_end = i_end < a.n_poses ? i_end : a.n_poses;
    const int q_first = a.ref_ptr[i_first < a.n_poses ? i_first : a.n_poses], q_end = a.ref_ptr[i_end > i_first ? i_end : i_first];
    int r0 = 0, r1 = 0;
    unsigned char pose_mode = 0;
    if (pose_on) {
        r0 = a.ref_ptr[i];
        r1 = a.ref_ptr[i + 1];
        pose_mode = a.pose_frozen[i];
    }
    const double mu_now = a.mu_dev ? *a.mu_dev : a.mu;
    __shared__ SolveDatasetDev s_ds[kSchurLdsDatasets];
    const bool ds_lds = a.n_ds > 0 && a.n_ds <= kSchurLdsDatasets;
    if (ds_lds && (int)threadIdx.x < a.n_ds) s_ds[threadIdx.x] = a.ds[threadIdx.x];   // (straight to LDS: no registers held across the kernel)
    if (gate_closed(a.gate, a.gate_expect)) return;
    // poses of this workgroup whose damped block was not positive definite: the last entry of the workgroup's partial, so
    // that the fixed-order sum over the workgroups delivers the count next to the Gram (no atomic, no counter to clear)
    __shared__ int s_bad;
    if (threadIdx.x == 0) {
        s_bad = 0;
        if (a.zero_u64 && blockIdx.x == 0) *a.zero_u64 = 0ull;
    }
    __syncthreads();
    // ---- the references of the workgroup's poses (a contiguous run of the CSR lists: the poses are consecutive), resolved ONCE
    // per workgroup into LDS: [block address | W | pose column | dataset].  Every gather below then reads addresses from LDS
    // and its global loads no longer hang off a chain pose -> reference -> dataset descriptor -> block, repeated per reference
    // and per lane (the rig: four references per pose, ~12 dependent round trips per lane).
    double *sm_V = sm_rows + (size_t)batches * poses_per_wg * 6 * CS;   // [batches * poses_per_wg][28]: V (21) | g (6) | has refs
    struct RefMeta {
        const double *Gb;
        int W, o, d, pad;
    };
    RefMeta *sm_ref = reinterpret_cast<RefMeta *>(sm_V + (size_t)batches * poses_per_wg * 28);   // [kSchurMaxRefs]
    const bool meta_lds = shared_gather && q_end - q_first <= kSchurMaxRefs;
    if (meta_lds) {
        for (int t = threadIdx.x; t < q_end - q_first; t += blockDim.x) {
            const int d = a.ref_ds[q_first + t];
            SolveDatasetDev D;
            if (ds_lds) D = s_ds[d];
            else D = a.ds[d];
            RefMeta m;
            m.Gb = D.gram + (size_t)a.ref_blk[q_first + t] * D.W * D.W;
            m.W = D.W;
            m.o = D.pose_off;
            m.d = d;
            m.pad = 0;
            sm_ref[t] = m;
        }
        __syncthreads();
    }
    auto ref_of = [&](int q, const double *&Gb, int &W, int &o, int &d) {
        if (meta_lds) {
            const RefMeta m = sm_ref[q - q_first];
            Gb = m.Gb;
            W = m.W;
            o = m.o;
            d = m.d;
        } else {
            d = a.ref_ds[q];
            const SolveDatasetDev D = a.ds[d];
            Gb = D.gram + (size_t)a.ref_blk[q] * D.W * D.W;
            W = D.W;
            o = D.pose_off;
        }
    };
    // ---- this lane's column of W_i^T (six values per reference), requested BEFORE the pose's V_i / g_i are shared: the
    // two gathers are independent and their loads are in flight together
    double w[6] = {0., 0., 0., 0., 0., 0.};
    if (pose_on && gcol < a.G) {
        for (int q = r0; q < r1; q++) {
            const double *Gb;
            int W, o, d;
            ref_of(q, Gb, W, o, d);
            const int lc = a.inv[d * a.G + gcol];
            if (lc < 0) continue;
#pragma unroll
            for (int c = 0; c < 6; c++) w[c] += Gb[lc * W + o + c];
        }
    }
    // ---- V_i and g_i of the pose, gathered ONCE per pose: lane e of the pose's C lanes adds entry e (of 27: the packed lower
    // triangle, then g) over the pose's references in order and parks it in LDS; every lane then reads the 27 values back.
    // (Each of the C lanes used to walk the whole gather itself -- 27 loads per reference and lane.)  Same order of additions:
    // same bits.
    if (shared_gather) {
        if (pose_on) {
            double *Vs = sm_V + (size_t)(bt * poses_per_wg + pl) * 28;
            for (int e = gcol; e < 27; e += C) {
                int r, c;   // entry e: (r, c) of the packed lower triangle, or (e - 21, residual column)
                if (e < 21) {
                    r = 0;
#pragma unroll
                    for (int k = 1; k < 6; k++) r += e >= k * (k + 1) / 2 ? 1 : 0;
                    c = e - r * (r + 1) / 2;
                } else {
                    r = e - 21;
                    c = -1;
                }
                double v = 0.;
                for (int q = r0; q < r1; q++) {
                    const double *Gb;
                    int W, o, d;
                    ref_of(q, Gb, W, o, d);
                    v += Gb[(o + r) * W + (c < 0 ? W - 1 : o + c)];
                }
                Vs[e] = v;
            }
            if (gcol == 0) Vs[27] = r1 > r0 ? 1. : 0.;
        }
        __syncthreads();
    }
    {
        double y[6] = {0., 0., 0., 0., 0., 0.};
        if (pose_on) {
            PoseFactor f;
            if (shared_gather) {
                const double *Vs = sm_V + (size_t)(bt * poses_per_wg + pl) * 28;
                double V[21];
#pragma unroll
                for (int k = 0; k < 21; k++) V[k] = Vs[k];
#pragma unroll
                for (int k = 0; k < 6; k++) f.gp[k] = Vs[21 + k];
                pose_factor_from(a, pose_mode, mu_now, V, Vs[27] != 0., f);
            } else {
                pose_factor(a, i, f);
            }
            if (gcol >= a.G) {
#pragma unroll
                for (int c = 0; c < 6; c++) w[c] = f.gp[c];
                double *rec = a.rec + (size_t)i * kPoseRec;
#pragma unroll
                for (int k = 0; k < 21; k++) rec[k] = f.L[k];
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    rec[21 + k] = f.gp[k];
                    rec[27 + k] = f.vd[k];
                }
                rec[33] = f.active ? 1. : 0.;
                if (f.mode == 0 && !f.pd && r1 > r0) atomicAdd(&s_bad, 1);
            }
            fwd6(f.L, w, y);
            double *out = a.rows + (size_t)i * 6 * C + gcol;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                y[k] = f.active ? y[k] : 0.;
                out[k * C] = y[k];
            }
        }
        if (lane_on && bt < batches) {
#pragma unroll
            for (int k = 0; k < 6; k++) sm_rows[(size_t)((bt * poses_per_wg + pl) * 6 + k) * CS + gcol] = y[k];
        }
    }
    __syncthreads();
    // entry-parallel Gram of the workgroup's rows (rows in increasing order: a fixed order)
    const int n_rows_wg = batches * poses_per_wg * 6, E = C * (C + 1) / 2;
    double *P = partials + (size_t)blockIdx.x * (C * C + 1);
    if (threadIdx.x == 0) P[C * C] = (double)s_bad;
    // Narrow systems (28 entries at G = 6) leave most of the workgroup idle while each entry's thread walks all rows -- 216
    // dependent multiply-adds at ~25 cycles: nch = 2, 4 or 8 lanes share an entry (rows nch apart) and add their parts in a
    // fixed butterfly order.
    int nch = 1;
    while (nch < 8 && E * nch * 2 <= (int)blockDim.x) nch *= 2;
    if (nch == 1) {   // wide systems: an entry per thread (and more), as before
        for (int e = threadIdx.x; e < E; e += blockDim.x) {
            int r = 0, rem = e;
            while (rem >= C - r) {
                rem -= C - r;
                r++;
            }
            const int c = r + rem;
            double s = 0.;
            for (int row = 0; row < n_rows_wg; row++) s += sm_rows[(size_t)row * CS + r] * sm_rows[(size_t)row * CS + c];
            P[r * C + c] = s;
            P[c * C + r] = s;
        }
        return;
    }
    for (int t = threadIdx.x; t < ((E * nch + kWave - 1) / kWave) * kWave; t += blockDim.x) {   // whole waves: the butterfly needs every lane
        const int e = t / nch, ch = t - e * nch;
        const bool on = e < E;
        int r = 0, rem = on ? e : 0;
        while (rem >= C - r) {
            rem -= C - r;
            r++;
        }
        const int c = r + rem;
        double s = 0.;
        if (on)
            for (int row = ch; row < n_rows_wg; row += nch) s += sm_rows[(size_t)row * CS + r] * sm_rows[(size_t)row * CS + c];
        for (int off = 1; off < nch; off <<= 1) s += __shfl_xor(s, off, kWave);
        if (on && ch == 0) {
            P[r * C + c] = s;
            P[c * C + r] = s;
        }
    }
}
#endif

// ceres::SoftLOneLoss(a) on one residual block = one image: rho(s) = 2 a^2 (sqrt(1 + s / a^2) - 1), s = r^T r.
// rho'' < 0, so Ceres' Corrector only scales residuals and Jacobian rows by sqrt(rho'): the block's Gram matrix
// becomes rho' * G, and its last entry (r^T r, twice the cost term) becomes rho(s).  One wave per block; the wave
// reads s before any of its lanes writes.
#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(64) void vg_gram_soft_l1_kernel(double *__restrict__ gram, int entries, double a2)
{
    double *G = gram + (size_t)blockIdx.x * entries;
    const double s = G[entries - 1];
    const double q = sqrt(1. + s / a2);
    const double w = 1. / q, rho = 2. * a2 * (q - 1.);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = threadIdx.x; e < entries; e += kWave) G[e] = e == entries - 1 ? rho : w * G[e];
}
#endif

struct BacksubArgs {
    SchurArgs s;
    const double *dg;              // [G] global step
    const long long *pose_param;   // [n_poses] offset of the pose's 6-vector in the parameter vector
    const long long *gcol_param;   // [G] offset of each global column in the parameter vector
    double *delta;                 // parameter-layout step vector
    double *scal;                  // [n_workgroups][5] partial sums: gp.dp | sum D dp^2 | |dp|^2 | |gp|^2 | |x_pose|^2
    unsigned long long *gmax_bits; // max |gp| over active poses, as the bit pattern of a non-negative double
    const double *x;               // current parameter vector
    double *xg;                    // [G] current values of the global columns (gathered for the host)
    const double *lo, *hi;         // [G] box of every global column (poses are unbounded)
    double *x_new;                 // clamp(x + delta, lo, hi) of the global columns and of this kernel's poses
    // The frames of the CANDIDATE point, built here (no chain-prep launch in front of the candidate's
    // evaluation): fold[d] describes dataset d as vg_chain_prep_multi_kernel sees it (count = 0: its Gram kernel walks the chain
    // itself, nothing to build), fold_gcol[d * kMaxChain + l] is the first global column of chain member l (-1: the sequence
    // member -- the pose at hand; constant transforms are columns too, with a zero step).  NULL: the caller launches the chain
    // prep as before.
    const PrepDataset *fold = nullptr;
    const int *fold_gcol = nullptr;
};

// dp_i = -V_i'^-1 (g_i + W_i^T dg) = -L^-T (L^-1 g_i + (L^-1 W_i^T) dg): everything needed is already in the
// six rows of the pose (last column = L^-1 g_i), no second gather from the Gram blocks.
// 16 lanes per pose: the six G-long dot products are read coalesced (16 consecutive doubles per row and step) and
// reduced inside the 16-lane group; the group's first lane finishes the 6 x 6 triangular solve.  One lane per pose
// (a serial loop over 6 G uncoalesced loads) took 31 us for 5 000 poses x 45 columns.  The per-pose terms of the model
// decrease / norms are reduced per workgroup in a fixed order, one partial per kBsPosesPerBlock poses.
constexpr int kBsGroup = 16;
constexpr int kBsThreads = 256;
constexpr int kBsPosesPerGroup = 2;
constexpr int kBsPosesPerBlock = (kBsThreads / kBsGroup) * kBsPosesPerGroup;

// dgv: the global step, in global memory (b.dg) or in the workgroup's LDS (the kernel that solves the reduced system itself)
// kFrames: the instantiation that also builds the candidate's frames (b.fold != NULL).  A template parameter because the chain
// walk is what sets the kernel's register count (202: two waves per SIMD); without it a mono problem's launch keeps more waves
// resident -- at 100 k poses the launch is bound by how many workgroups a CU holds, not by their latency.
template <int kJ, bool kFrames>
__device__ __forceinline__ void backsub_body(const BacksubArgs &b, const double *dgv)
{
    const SchurArgs &a = b.s;
    const int t = blockIdx.x * kBsThreads + threadIdx.x;
    if (t < a.G) {
        const long long gp = b.gcol_param[t];
        b.delta[gp] = dgv[t];
        b.xg[t] = b.x[gp];
        b.x_new[gp] = clampd(b.x[gp] + dgv[t], b.lo[t], b.hi[t]);
    }
    // candidate frames: every workgroup needs the candidate values of the global columns (the expression above: same bits)
    __shared__ double s_gx[kBsThreads / 2];
    static_assert(kBsThreads / 2 >= 127, "a slot per global column");
    if (kFrames && b.fold) {
        if ((int)threadIdx.x < a.G) {
            const long long gp = b.gcol_param[threadIdx.x];
            s_gx[threadIdx.x] = clampd(b.x[gp] + dgv[threadIdx.x], b.lo[threadIdx.x], b.hi[threadIdx.x]);
        }
        __syncthreads();
    }
    if ((int)(blockIdx.x * kBsPosesPerBlock) >= a.n_poses) return;  // workgroups that only carry global columns
    const int gl = threadIdx.x & (kBsGroup - 1), grp = threadIdx.x >> 4;
    const int C = a.G + 1;
    double s0 = 0., s1 = 0., s2 = 0., s3 = 0., s4 = 0., s5 = 0.;
    double xc[kBsPosesPerGroup][6] = {};   // the group's candidate poses (valid in its first lane)
    int ref0[kBsPosesPerGroup], ref1[kBsPosesPerGroup];
#pragma unroll
    for (int q = 0; q < kBsPosesPerGroup; q++) {
        const int i = blockIdx.x * kBsPosesPerBlock + q * (kBsThreads / kBsGroup) + grp;
        const bool pv = i < a.n_poses;  // uniform inside the 16-lane group
        const int ii = pv ? i : 0;
        const double *rows = a.rows + (size_t)ii * 6 * C;
        // everything the group's first lane needs later is loaded by all 16 lanes up front (same addresses: one
        // transaction per group), so the loads overlap the dot products instead of forming a second and third
        // round trip behind the branch
        const double *rec = a.rec + (size_t)ii * kPoseRec;
        const long long pp = b.pose_param[ii];
        ref0[q] = (kFrames && b.fold && pv) ? a.ref_ptr[ii] : 0;
        ref1[q] = (kFrames && b.fold && pv) ? a.ref_ptr[ii + 1] : 0;
        double L[21], gk[6], dk[6], xk[6], y[6], yl[6];
#pragma unroll
        for (int k = 0; k < 21; k++) L[k] = rec[k];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            gk[k] = rec[21 + k];
            dk[k] = rec[27 + k];
            xk[k] = b.x[pp + k];
            yl[k] = rows[k * C + a.G];
        }
        const bool active = rec[33] != 0.;
        // G <= 16 kJ - 1: at most kJ columns per lane and row (kJ = 4 up to 63 global columns, 8 up to 127); all
        // 6 kJ loads are issued before the first use (a loop over a run-time G waits for every load in turn)
        double dgv_l[kJ], rv[6][kJ];
#pragma unroll
        for (int j = 0; j < kJ; j++) {
            const int g = gl + kBsGroup * j;
            dgv_l[j] = g < a.G ? dgv[g] : 0.;
#pragma unroll
            for (int k = 0; k < 6; k++) rv[k][j] = g < a.G ? rows[k * C + g] : 0.;
        }
#pragma unroll
        for (int k = 0; k < 6; k++) {
            double s = 0.;
#pragma unroll
            for (int j = 0; j < kJ; j++) s += rv[k][j] * dgv_l[j];
            y[k] = s;
        }
#pragma unroll
        for (int k = 0; k < 6; k++)
#pragma unroll
            for (int off = kBsGroup / 2; off >= 1; off >>= 1) y[k] += __shfl_xor(y[k], off, kWave);
        if (gl == 0 && pv) {
            double x[6];
#pragma unroll
            for (int k = 0; k < 6; k++) y[k] += yl[k];
            bwd6(L, y, x);
            double *dp = b.delta + pp;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const double v = active ? -x[c] : 0.;
                dp[c] = v;
                b.x_new[pp + c] = xk[c] + v;  // the candidate point (poses are unbounded)
                xc[q][c] = xk[c] + v;
                const double g = gk[c];
                s0 += g * v;
                s1 += clampd(dk[c], a.dmin, a.dmax) * v * v;
                if (active) s2 = fmax(s2, fabs(g));
                if (active) s4 += g * g;
                s3 += v * v;
                s5 += xk[c] * xk[c];
            }
        }
    }
    if (kFrames && b.fold) {
        // The frames of the group's two poses at the candidate point, for every dataset block that refers to them: lanes 0..7 of
        // the group take the references of the first pose, lanes 8..15 those of the second (a rig: four each), all chains of the
        // workgroup side by side -- the walk is one dependent chain per lane, exactly what vg_chain_prep_multi_kernel runs one
        // launch later with most of the chip idle.
        static_assert(kBsPosesPerGroup == 2 && kBsGroup == 16, "two poses per 16-lane group");
        const int half = gl >> 3, hl = gl & 7;
        double xp[6];
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const double v0 = __shfl(xc[0][c], 0, kBsGroup), v1 = __shfl(xc[1][c], 0, kBsGroup);
            xp[c] = half ? v1 : v0;
        }
        const int r_end = half ? ref1[1] : ref1[0];
        for (int r = (half ? ref0[1] : ref0[0]) + hl; r < r_end; r += 8) {
            const int ds = a.ref_ds[r];
            const PrepDataset *D = b.fold + ds;
            if (D->count == 0) continue;
            const long long blk = a.ref_blk[r];
            const int *gc = b.fold_gcol + ds * kMaxChain;
            build_frame_vals(D->chain.L, D->chain.status,
                             [&](int l, double *xi) {
                                 const int g0 = gc[l];
#pragma unroll
                                 for (int c = 0; c < 6; c++) xi[c] = g0 < 0 ? xp[c] : s_gx[g0 + c];
                             },
                             D->frames + blk * D->frame_stride_d);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s0 += __shfl_xor(s0, off, kWave);
        s1 += __shfl_xor(s1, off, kWave);
        s3 += __shfl_xor(s3, off, kWave);
        s4 += __shfl_xor(s4, off, kWave);
        s5 += __shfl_xor(s5, off, kWave);
        s2 = fmax(s2, __shfl_xor(s2, off, kWave));
    }
    __shared__ double red[kBsThreads / kWave][6];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s3; red[wave][3] = s4; red[wave][4] = s5; red[wave][5] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *sc = b.scal + (size_t)blockIdx.x * 5;
        double m = 0.;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            double v = 0.;
#pragma unroll
            for (int w = 0; w < kBsThreads / kWave; w++) v += red[w][k];  // fixed order
            sc[k] = v;
        }
#pragma unroll
        for (int w = 0; w < kBsThreads / kWave; w++) m = fmax(m, red[w][5]);
        atomicMax(b.gmax_bits, (unsigned long long)__double_as_longlong(m));  // order preserving for m >= 0
    }
}

template <int kJ, bool kFrames>
__global__ __launch_bounds__(kBsThreads) void vg_backsub_kernel(BacksubArgs b)
{
    if (gate_closed(b.s.gate, b.s.gate_expect)) return;
    backsub_body<kJ, kFrames>(b, b.dg);
}

// kJ = columns per lane of a pose's 16-lane group: the narrowest instantiation that covers G columns (registers: 202 at kJ = 4,
// two waves per SIMD; a mono problem needs one column per lane)
inline void launch_backsub(hipStream_t st, int G, unsigned int grid, const BacksubArgs &ba)
{
    const bool fr = ba.fold != nullptr;
    if (G < 16) {
        if (fr) hipLaunchKernelGGL((vg_backsub_kernel<1, true>), dim3(grid), dim3(kBsThreads), 0, st, ba);
        else hipLaunchKernelGGL((vg_backsub_kernel<1, false>), dim3(grid), dim3(kBsThreads), 0, st, ba);
    } else if (G < 32) {
        if (fr) hipLaunchKernelGGL((vg_backsub_kernel<2, true>), dim3(grid), dim3(kBsThreads), 0, st, ba);
        else hipLaunchKernelGGL((vg_backsub_kernel<2, false>), dim3(grid), dim3(kBsThreads), 0, st, ba);
    } else if (G < 64) {
        if (fr) hipLaunchKernelGGL((vg_backsub_kernel<4, true>), dim3(grid), dim3(kBsThreads), 0, st, ba);
        else hipLaunchKernelGGL((vg_backsub_kernel<4, false>), dim3(grid), dim3(kBsThreads), 0, st, ba);
    } else {
        if (fr) hipLaunchKernelGGL((vg_backsub_kernel<8, true>), dim3(grid), dim3(kBsThreads), 0, st, ba);
        else hipLaunchKernelGGL((vg_backsub_kernel<8, false>), dim3(grid), dim3(kBsThreads), 0, st, ba);
    }
}

// x_new = x + delta (pose parameters: unbounded)
#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_apply_step_kernel(const double *__restrict__ x, const double *__restrict__ delta, long long n,
                                                             double *__restrict__ x_new)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    x_new[i] = x[i] + delta[i];
}
#endif

// Gram of a plain row-major matrix X [n_rows][C], one wave per group of `rows_per_group` rows (a multiple of 4),
// 4 rows per MFMA in increasing order: out[group][C*C].  Groups are then added by the slab / final sum kernels.
template <int T>
__global__ __launch_bounds__(256) void vg_dense_gram_kernel(const double *__restrict__ X, unsigned int n_rows, int C,
                                                             unsigned int rows_per_group, unsigned int n_groups,
                                                             double *__restrict__ out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const unsigned int grp = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (grp >= n_groups) return;
    const int c = lane & 15, k = lane >> 4;
    const unsigned int row0 = grp * rows_per_group;
    f64x4 acc[T][T];
    zero_acc<T>(acc);
    for (unsigned int t = 0; t < rows_per_group / 4; t++) {
        const unsigned int row = row0 + 4 * t + k;
        double v[T];
#pragma unroll
        for (int j = 0; j < T; j++) {
            const int col = 16 * j + c;
            v[j] = (col < C && row < n_rows) ? X[(size_t)row * C + col] : 0.;
        }
#pragma unroll
        for (int ti = 0; ti < T; ti++)
#pragma unroll
            for (int tj = ti; tj < T; tj++) acc[ti][tj] = mfma_f64_16x16x4(v[ti], v[tj], acc[ti][tj]);
    }
    store_gram<T>(acc, out + (size_t)grp * C * C, C, lane);
}

// The same Gram for wide matrices (C > 64, i.e. more than 63 global columns): one wave per (row group, tile pair),
// blockIdx.y enumerates the upper-triangle tile pairs; every pair writes its own entries of out[group].
#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_dense_gram_pair_kernel(const double *__restrict__ X, unsigned int n_rows, int C,
                                                                  unsigned int rows_per_group, unsigned int n_groups,
                                                                  double *__restrict__ out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const unsigned int grp = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (grp >= n_groups) return;
    // pair index -> (ti <= tj)
    int ti = 0, rest = (int)blockIdx.y, T = (C + 15) / 16;
    while (rest >= T - ti) {
        rest -= T - ti;
        ti++;
    }
    const int tj = ti + rest;
    const int c = lane & 15, k = lane >> 4;
    const unsigned int row0 = grp * rows_per_group;
    f64x4 acc = {0., 0., 0., 0.};
    for (unsigned int t = 0; t < rows_per_group / 4; t++) {
        const unsigned int row = row0 + 4 * t + k;
        const int ci = 16 * ti + c, cj = 16 * tj + c;
        const double vi = (ci < C && row < n_rows) ? X[(size_t)row * C + ci] : 0.;
        const double vj = (cj < C && row < n_rows) ? X[(size_t)row * C + cj] : 0.;
        acc = mfma_f64_16x16x4(vi, vj, acc);
    }
    double *g = out + (size_t)grp * C * C;
    const int row0l = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int i = 16 * ti + row0l + 4 * r, j = 16 * tj + c;
        if (i < C && j < C) {
            g[i * C + j] = acc[r];
            if (ti != tj) g[j * C + i] = acc[r];
        }
    }
}
#endif

}  // namespace vg
