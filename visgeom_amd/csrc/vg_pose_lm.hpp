// vg_pose_lm.hpp -- the per-image pose refinement of estimateInitialGrid
// (src/calibration/unified_calibration.cpp:1137-1155): ONE independent 6-DOF problem per image -- one
// GenericProjectionJac block with chain {DIRECT}, SoftLOneLoss(25), intrinsics constant, at most 500 iterations of
// Ceres' default trust-region Levenberg-Marquardt.  The reference runs them one after the other (10 k tiny Ceres solves
// at the benchmark scale); here every image is a half-wave (32 lanes, 3 corners per lane on an 8 x 12 board) that runs
// its WHOLE solve inside one launch: own trust-region radius, own step acceptance, own convergence tests -- an image
// with failed projections (1e15 residuals) or a bad start cannot touch its neighbours (the first version
// refined all images in one joint problem with one shared radius).  The grid is persistent (round 5): a half-wave that is
// done with its image takes the next one from a counter.
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
    unsigned int *next;   // work counter, zero at launch: images beyond the grid's first 8 per workgroup are handed out through it
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

// [J | r]^T [J | r] of one image at pose x (all 32 lanes of the half-wave hold x): the 28 entries (upper triangle, row major) land
// in `gdst` (LDS, this half-wave's), visible to all its lanes on return.  fr: this half-wave's frame scratch (kFrameDoubles).
template <int MODEL>
__device__ __forceinline__ void pose_gram(const PoseLmArgs &a, unsigned int b, int sl, const double (&x)[6], double *fr, double *gdst)
{
    constexpr int K = CameraTraits<MODEL>::K;
    using d2 = HIP_vector_type<double, 2>;
#ifdef VG_POSE_REFERENCE_WALK   // A/B library: the reference-order walk
    if (sl == 0) build_frame_single_direct(x, fr);
#else
    if (sl == 0) build_frame_single_direct_fast(x, fr);   // 31 lanes wait for this one: the short walk (vg_geometry.hpp), the frame to 1e-16
#endif
    wave_lds_fence();
    double acc[kPoseE];
#pragma unroll
    for (int e = 0; e < kPoseE; e++) acc[e] = 0.;
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
#pragma unroll
        for (int r = 0, q = 0; r < kPoseW; r++)
#pragma unroll
            for (int cc = r; cc < kPoseW; cc++, q++) acc[q] += rw[0][r] * rw[0][cc] + rw[1][r] * rw[1][cc];
    }
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
    if (real > 0) gdst[base] = s5[0];
    wave_lds_fence();
}

__device__ __forceinline__ int tri7(int r, int c) { return r * kPoseW - r * (r - 1) / 2 + (c - r); }  // r <= c

#ifndef VG_POSE_WAVES
#define VG_POSE_WAVES 2   // waves per SIMD the register allocation aims at
#endif

// One half-wave per image AT A TIME, 8 half-waves per 256-thread workgroup, a PERSISTENT grid that fills the device once: a
// half-wave whose image has finished takes the next one from a counter (a.next) while its sibling half carries on with its own --
// images need 3 to 46 iterations (mean 5), and in round 4's one-image-per-half-wave launch every wave, and every round of
// workgroups, lasted as long as its slowest image.  The two Gram sets of an image (current point, candidate) live in LDS, not in
// registers: 340 -> under 256 registers, two waves per SIMD.  What an image computes does not depend on where or next to whom it runs.
template <int MODEL>
__global__ __launch_bounds__(kValuThreads, VG_POSE_WAVES) void vg_pose_lm_kernel(PoseLmArgs a)
{
    constexpr int FS = frame_stride(1), kScratch = FS + 2 * kPoseE;
    __shared__ __attribute__((aligned(16))) double lds[kValuImagesPerBlock * kScratch];
    const int tid = threadIdx.x, sl = tid & (kValuLanesPerImage - 1), lane = tid & (kWave - 1);
    double *fr = lds + (tid / kValuLanesPerImage) * kScratch, *gbuf = fr + FS;

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

    unsigned int b = blockIdx.x * kValuImagesPerBlock + (unsigned)(tid / kValuLanesPerImage);   // the first image is this half-wave's by position
    bool has = b < a.n_images, fresh = true;
    int cur = 0, it = 0, iters = 0, term = VG_TERM_NO_CONVERGENCE;
    double x[6] = {0., 0., 0., 0., 0., 0.}, w_x = 1., cost = 0., radius = a.radius0, decrease_factor = 2.;
    if (has) {
#pragma unroll
        for (int k = 0; k < 6; k++) x[k] = a.poses[(size_t)b * 6 + k];
    }

    while (__builtin_amdgcn_ballot_w64(has)) {
        // ---- the point to evaluate: the image's starting point (fresh), or the LM step from the current Gram set
        double xc[6], gk[6], Dk[6], dx[6];
        double mu = 0., gdx = 0., ddx = 0., dx2 = 0., x2 = 0., gmax = 0.;
        bool pd = true;
#pragma unroll
        for (int k = 0; k < 6; k++) xc[k] = x[k];
        if (has && !fresh) {
            const double *G = gbuf + cur * kPoseE;
            mu = 1. / radius;
            // (rho' J^T J + mu D) delta = -rho' J^T r, D = clamp(diag(rho' J^T J))   -- Ceres' LM on the corrected block
            double A[21], L[21], y[6];
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
        }
        double *gcand = gbuf + (1 - cur) * kPoseE;
        pose_gram<MODEL>(a, has ? b : 0u, sl, xc, fr, gcand);   // both halves of the wave, whatever their state
        if (!has) continue;
        bool done = false;
        if (fresh) {   // the evaluation at the starting point
            double rho_x;
            robust(gcand[kPoseE - 1], rho_x, w_x);
            cost = 0.5 * rho_x;
            cur = 1 - cur;
            fresh = false;
            done = a.max_iter < 1;
        } else {
            it++;
            iters = it;
            double rho_c, w_c;
            robust(gcand[kPoseE - 1], rho_c, w_c);
            const double cost_c = 0.5 * rho_c;
            const double model_change = 0.5 * (mu * ddx - gdx);  // 1/2 delta^T (mu D delta - g)
            const double cost_change = cost - cost_c;
            const double gain = (pd && model_change > 0.) ? cost_change / model_change : -1.;
            if (pd && gmax <= a.gtol) {
                term = VG_TERM_CONVERGENCE_GRADIENT;
                done = true;
            } else if (pd && sqrt(dx2) <= a.ptol * (sqrt(x2) + a.ptol)) {
                term = VG_TERM_CONVERGENCE_PARAMETER;
                done = true;
            } else if (pd && isfinite(cost_c) && gain > a.min_rel_decrease) {
                // (Unlike vg_lm_accept_kernel the function tolerance is tested AFTER a step has been accepted: Ceres tests it before
                //  it decides about the step and returns at the point in front of it -- with this solve's tolerance of 1e-6 that leaves
                //  the pose up to ~1e-5 short of the per-image optimum the tests hold it to; the accepted last step is at least as good
                //  a seed for the global solve.)
#pragma unroll
                for (int k = 0; k < 6; k++) x[k] = xc[k];
                cur = 1 - cur;
                w_x = w_c;
                const double prev = cost;
                cost = cost_c;
                const double f = 1. - (2. * gain - 1.) * (2. * gain - 1.) * (2. * gain - 1.);
                radius = fmin(radius / fmax(f, 1. / 3.), a.max_radius);
                decrease_factor = 2.;
                if (fabs(prev - cost) <= a.ftol * prev) {
                    term = VG_TERM_CONVERGENCE_FUNCTION;
                    done = true;
                }
            } else {
                radius /= decrease_factor;
                decrease_factor *= 2.;
                if (radius < a.min_radius) {
                    term = VG_TERM_RADIUS_TOO_SMALL;
                    done = true;
                }
            }
            if (it >= a.max_iter) done = true;   // (term stays NO_CONVERGENCE unless a test fired in this iteration)
        }
        if (done) {
            unsigned int nb = 0;
            if (sl == 0) {
#pragma unroll
                for (int k = 0; k < 6; k++) a.poses[(size_t)b * 6 + k] = x[k];
                if (a.iterations) a.iterations[b] = iters;
                if (a.final_cost) a.final_cost[b] = cost;
                if (a.termination) a.termination[b] = term;
                nb = gridDim.x * kValuImagesPerBlock + atomicAdd(a.next, 1u);   // the next image nobody has taken
            }
            b = (unsigned int)__shfl((int)nb, lane & kValuLanesPerImage, kWave);
            has = b < a.n_images;
            fresh = true;
            it = iters = 0;
            term = VG_TERM_NO_CONVERGENCE;
            radius = a.radius0;
            decrease_factor = 2.;
            w_x = 1.;
            if (has) {
#pragma unroll
                for (int k = 0; k < 6; k++) x[k] = a.poses[(size_t)b * 6 + k];
            }
        }
    }
}

}  // namespace vg
