// vg_pose_lm.hpp -- the per-image pose refinement of estimateInitialGrid
// (src/calibration/unified_calibration.cpp:1137-1155): ONE independent 6-DOF problem per image -- one
// GenericProjectionJac block with chain {DIRECT}, SoftLOneLoss(25), intrinsics constant, at most 500 iterations of
// Ceres' default trust-region Levenberg-Marquardt.  The reference runs them one after the other (10 k tiny Ceres solves
// at the benchmark scale); here every image is a half-wave (32 lanes, 3 corners per lane on an 8 x 12 board) that runs
// its WHOLE solve inside one launch: own trust-region radius, own step acceptance, own convergence tests -- an image
// with failed projections (1e15 residuals) or a bad start cannot touch its neighbours (the first version
// refined all images in one joint problem with one shared radius).
//
// Per iteration and image: chain walk (lane 0 of the half-wave, reference order), the 2 x 6 pose rows and the residual
// pair of every corner (reference-order device functions of vg_camera.hpp), the 7 x 7 Gram [J | r]^T [J | r] summed over
// the 32 lanes with the select-free halving exchanges of vg_gram_valu.hpp, a 6 x 6 Cholesky solve executed by every
// lane (the instruction stream is the same for one lane or 32), candidate evaluation, Ceres' acceptance / radius rule.
// With ONE residual block per problem the loss function rho(s), s = |r|^2, is a monotone function of the plain cost:
// the minimiser is that of the unweighted problem; rho only enters the gain ratio and the convergence tests
// (Ceres' Corrector with rho'' < 0: residuals and Jacobian scaled by sqrt(rho')).
#pragma once

#include "vg_gram_valu.hpp"
#include "vg_solver.hpp"

namespace vg {

struct PoseLmArgs {
    const double *board;  // [N][3]
    const double *obs;    // [n_images][N][2]
    const double *intr;   // [K]
    double *poses;        // [n_images][6]  in: start, out: result
    int *iterations;      // [n_images] or NULL
    double *final_cost;   // [n_images] or NULL  (rho(s) / 2, as Ceres reports it)
    int *termination;     // [n_images] or NULL  (vg_termination)
    unsigned int n_images, N;
    int max_iter;
    double a2;            // SoftLOneLoss scale squared, 0 = no loss function
    double ftol, gtol, ptol, radius0, max_radius, min_radius, min_rel_decrease, dmin, dmax;
};

constexpr int kPoseW = 7, kPoseE = kPoseW * (kPoseW + 1) / 2;  // 6 pose columns + residual: 28 entries

template <int N, int LEVEL>
__device__ __forceinline__ void halve_level(const double (&v)[N], double (&o)[(N + 1) / 2], bool bit)
{
    constexpr int H = (N + 1) / 2;
#pragma unroll
    for (int k = 0; k < H; k++) o[k] = halve_pair<level_dist(LEVEL)>(v[k], (k + H < N) ? v[k + H] : 0., bit);
}

// [J | r]^T [J | r] of one image at pose x (all 32 lanes hold x); every lane returns all 28 entries (upper triangle,
// row major).  lds: this half-wave's scratch, kFrameDoubles + kPoseE doubles.
template <int MODEL>
__device__ __forceinline__ void pose_gram(const PoseLmArgs &a, unsigned int b, int sl, const double (&x)[6], double *lds,
                                          double (&G)[kPoseE], bool &any_failed)
{
    constexpr int K = CameraTraits<MODEL>::K, FS = frame_stride(1);
    using d2 = HIP_vector_type<double, 2>;
    double *fr = lds, *gl = lds + FS;
    if (sl == 0) build_frame_single_direct(x, fr);
    wave_lds_fence();
    double acc[kPoseE];
#pragma unroll
    for (int e = 0; e < kPoseE; e++) acc[e] = 0.;
    bool failed = false;
    for (unsigned int c = sl; c < a.N; c += kValuLanesPerImage) {
        const double g0 = a.board[3 * c], g1 = a.board[3 * c + 1], g2 = a.board[3 * c + 2];
        const d2 ob = reinterpret_cast<const d2 *>(a.obs)[(size_t)b * a.N + c];
        const double X0 = (fr[0] * g0 + fr[1] * g1 + fr[2] * g2) + fr[9];
        const double X1 = (fr[3] * g0 + fr[4] * g1 + fr[5] * g2) + fr[10];
        const double X2 = (fr[6] * g0 + fr[7] * g1 + fr[8] * g2) + fr[11];
        CornerEval<K> e;
        eval_corner<MODEL, true, false>(a.intr, X0, X1, X2, e);
        double rows[12], rw[2][kPoseW];
        pose_rows(e.P, X0, X1, X2, fr + 12, rows);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            rw[0][j] = rows[j];
            rw[1][j] = rows[6 + j];
        }
        rw[0][6] = e.ok ? e.u - ob.x : kDoubleBig;  // calib_cost_functions.cpp:66-70
        rw[1][6] = e.ok ? e.v - ob.y : kDoubleBig;
        failed = failed || !e.ok;
#pragma unroll
        for (int r = 0, q = 0; r < kPoseW; r++)
#pragma unroll
            for (int cc = r; cc < kPoseW; cc++, q++) acc[q] += rw[0][r] * rw[0][cc] + rw[1][r] * rw[1][cc];
    }
    any_failed = failed;
    // sum over the 32 lanes: 28 -> 14 -> 7 -> 4 -> 2 -> 1
    double s1[14], s2[7], s3[4], s4[2], s5[1];
    halve_level<28, 1>(acc, s1, sl & 16);
    halve_level<14, 2>(s1, s2, sl & 8);
    halve_level<7, 3>(s2, s3, sl & 4);
    halve_level<4, 4>(s3, s4, sl & 2);
    halve_level<2, 5>(s4, s5, sl & 1);
    int base = 0, real = kPoseE, n = kPoseE;
#pragma unroll
    for (int s = 0; s < 5; s++) {
        const int H = (n + 1) / 2;
        const bool bit = (sl >> (4 - s)) & 1;
        base += bit ? H : 0;
        real = bit ? (real - H > 0 ? real - H : 0) : (real < H ? real : H);
        n = H;
    }
    if (real > 0) gl[base] = s5[0];
    wave_lds_fence();
#pragma unroll
    for (int e = 0; e < kPoseE; e++) G[e] = gl[e];
    wave_lds_fence();  // the scratch is rewritten by the next call
}

__device__ __forceinline__ int tri7(int r, int c) { return r * kPoseW - r * (r - 1) / 2 + (c - r); }  // r <= c

// One half-wave per image, 8 images per 256-thread workgroup.
template <int MODEL>
__global__ __launch_bounds__(kValuThreads) void vg_pose_lm_kernel(PoseLmArgs a)
{
    constexpr int FS = frame_stride(1), kScratch = FS + kPoseE + 1;
    __shared__ __attribute__((aligned(16))) double lds[kValuImagesPerBlock * kScratch];
    const int tid = threadIdx.x, sl = tid & (kValuLanesPerImage - 1);
    const unsigned int b_raw = blockIdx.x * kValuImagesPerBlock + (unsigned)(tid / kValuLanesPerImage);
    const bool bvalid = b_raw < a.n_images;
    const unsigned int b = bvalid ? b_raw : a.n_images - 1;  // surplus half-waves shadow the last image, never store
    double *scratch = lds + (tid / kValuLanesPerImage) * kScratch;

    double x[6], xc[6], G[kPoseE], Gc[kPoseE];
#pragma unroll
    for (int k = 0; k < 6; k++) x[k] = a.poses[(size_t)b * 6 + k];
    bool fl;
    pose_gram<MODEL>(a, b, sl, x, scratch, G, fl);

    auto robust = [&](double s, double &rho, double &w) {  // SoftLOneLoss(a): rho(s) = 2 a^2 (sqrt(1 + s / a^2) - 1)
        if (a.a2 > 0.) {
            const double q = sqrt(1. + s / a.a2);
            rho = 2. * a.a2 * (q - 1.);
            w = 1. / q;
        } else {
            rho = s;
            w = 1.;
        }
    };
    double rho_x, w_x;
    robust(G[kPoseE - 1], rho_x, w_x);
    double cost = 0.5 * rho_x, radius = a.radius0, decrease_factor = 2.;
    int term = VG_TERM_NO_CONVERGENCE, iters = 0;
    bool active = true;

    for (int it = 1; it <= a.max_iter; it++) {
        if (!__builtin_amdgcn_ballot_w64(active)) break;
        const double mu = 1. / radius;
        // (rho' J^T J + mu D) delta = -rho' J^T r, D = clamp(diag(rho' J^T J))   -- Ceres' LM on the corrected block
        double A[21], gk[6], Dk[6], L[21], y[6], dx[6];
        bool pd = true;
#pragma unroll
        for (int r = 0; r < 6; r++) {
#pragma unroll
            for (int c = 0; c <= r; c++) A[tri(r, c)] = w_x * G[tri7(c, r)];
            gk[r] = w_x * G[tri7(r, 6)];
            Dk[r] = clampd(A[tri(r, r)], a.dmin, a.dmax);
            A[tri(r, r)] += mu * Dk[r];
        }
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c <= r; c++) {
                double s = A[tri(r, c)];
#pragma unroll
                for (int k = 0; k < c; k++) s -= L[tri(r, k)] * L[tri(c, k)];
                if (r == c) {
                    if (!(s > 0.)) { pd = false; s = 1.; }
                    L[tri(r, r)] = sqrt(s);
                } else {
                    L[tri(r, c)] = s / L[tri(c, c)];
                }
            }
        fwd6(L, gk, y);
        bwd6(L, y, dx);
        double gdx = 0., ddx = 0., dx2 = 0., x2 = 0., gmax = 0.;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            dx[k] = -dx[k];
            xc[k] = x[k] + dx[k];
            gdx += gk[k] * dx[k];
            ddx += Dk[k] * dx[k] * dx[k];
            dx2 += dx[k] * dx[k];
            x2 += x[k] * x[k];
            gmax = fmax(gmax, fabs(gk[k]));
        }
        bool flc;
        pose_gram<MODEL>(a, b, sl, xc, scratch, Gc, flc);  // both images of the wave, whatever their state
        if (!active) continue;
        iters = it;
        double rho_c, w_c;
        robust(Gc[kPoseE - 1], rho_c, w_c);
        const double cost_c = 0.5 * rho_c;
        const double model_change = 0.5 * (mu * ddx - gdx);  // 1/2 delta^T (mu D delta - g)
        const double cost_change = cost - cost_c;
        const double gain = (pd && model_change > 0.) ? cost_change / model_change : -1.;
        if (pd && gmax <= a.gtol) {
            term = VG_TERM_CONVERGENCE_GRADIENT;
            active = false;
            continue;
        }
        if (pd && sqrt(dx2) <= a.ptol * (sqrt(x2) + a.ptol)) {
            term = VG_TERM_CONVERGENCE_PARAMETER;
            active = false;
            continue;
        }
        // (Unlike vg_lm_accept_kernel the function tolerance is tested AFTER a step has been accepted: Ceres tests it before
        //  it decides about the step and returns at the point in front of it -- with this solve's tolerance of 1e-6 that leaves
        //  the pose up to ~1e-5 short of the per-image optimum the tests hold it to; the accepted last step is at least as good
        //  a seed for the global solve.)
        const bool success = pd && isfinite(cost_c) && gain > a.min_rel_decrease;
        if (success) {
#pragma unroll
            for (int k = 0; k < 6; k++) x[k] = xc[k];
#pragma unroll
            for (int e = 0; e < kPoseE; e++) G[e] = Gc[e];
            w_x = w_c;
            const double prev = cost;
            cost = cost_c;
            const double f = 1. - (2. * gain - 1.) * (2. * gain - 1.) * (2. * gain - 1.);
            radius = fmin(radius / fmax(f, 1. / 3.), a.max_radius);
            decrease_factor = 2.;
            if (fabs(prev - cost) <= a.ftol * prev) {
                term = VG_TERM_CONVERGENCE_FUNCTION;
                active = false;
            }
        } else {
            radius /= decrease_factor;
            decrease_factor *= 2.;
            if (radius < a.min_radius) {
                term = VG_TERM_RADIUS_TOO_SMALL;
                active = false;
            }
        }
    }
    if (bvalid && sl == 0) {
#pragma unroll
        for (int k = 0; k < 6; k++) a.poses[(size_t)b_raw * 6 + k] = x[k];
        if (a.iterations) a.iterations[b_raw] = iters;
        if (a.final_cost) a.final_cost[b_raw] = cost;
        if (a.termination) a.termination[b_raw] = term;
    }
}

}  // namespace vg
