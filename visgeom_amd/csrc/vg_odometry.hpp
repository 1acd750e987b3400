// vg_odometry.hpp -- host side of the OdometryPrior residual block of the calibration problem
// (struct OdometryPrior, include/calibration/calib_cost_functions.h:64-77; constructor
// src/calibration/calib_cost_functions.cpp:119-167, Evaluate :171-212).  Six residuals between two CONSECUTIVE
// elements of a sequence transform:  r = A * (zetaPrior^-1 o (xi1^-1 o xi2)),  with the reference's Jacobians
//     dr/dxi1 = -A * screwTransfInv(zeta) * blockdiag(R10, R10 M(r1)),   dr/dxi2 = A * blockdiag(R20, R20 M(r2)).
// "few rows, CPU is fine" -- these blocks are evaluated on the host; what they change is
// the structure of the pose system (block tridiagonal inside the sequence), handled in vg_solver_impl.hpp.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

#include "vg_transf_host.hpp"

namespace vgodo {

using vgth::Array6d;

struct Block {
    int tf = -1;        // sequence transform
    int64_t i = 0;      // couples elements i and i + 1
    double zeta[6];     // xi1_odom^-1 o xi2_odom
    double A[36];       // row-major
    // OdometryCost only (src/calibration/odometry_cost_function.cpp): the wheel increments of the interval and the
    // extra parameter block [radius_left, radius_right, track_gauge] the residual also depends on
    int pblock = -1;
    std::vector<double> dq;  // [n][2]
};

inline void mat3_mul(const double *A, const double *B, double *C) { vg::mat3_mul(A, B, C); }

inline void mat6_mul(const double *A, const double *B, double *C)
{
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double s = 0.;
            for (int k = 0; k < 6; k++) s += A[6 * i + k] * B[6 * k + j];
            C[6 * i + j] = s;
        }
}

// the weighting matrix both odometry blocks derive from their zetaPrior
// (calib_cost_functions.cpp:127-167 == odometry_cost_function.cpp:160-194)
inline void weight_from_zeta(Block &b, double errV, double errW, double lambda)
{
    const double MIN_SIGMA_V = 0.01, MIN_SIGMA_W = 0.01, MIN_DELTA = 0.01, MIN_L = 0.01;
    const double delta = std::fmax(vg::norm3(b.zeta + 3), MIN_DELTA);
    const double l = std::fmax(vg::norm3(b.zeta), MIN_L);
    const double delta2 = delta / 2., l2 = l / 2.;
    const double s = std::sin(delta2), co = std::cos(delta2);
    const double dfdu[3][2] = {{co, l2 * s}, {-s, l2 * co}, {0., 1.}};
    const double Cu0 = std::fmax(errV * errV * l * l, MIN_SIGMA_V * MIN_SIGMA_V);
    const double Cu1 = std::fmax(errW * errW * delta * delta, MIN_SIGMA_W * MIN_SIGMA_W);
    double Cx[9];
    for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++)
            Cx[3 * r + q] = dfdu[r][0] * Cu0 * dfdu[q][0] + dfdu[r][1] * Cu1 * dfdu[q][1] + (r == q ? lambda * lambda : 0.);
    // CxInv = Cx^-1 (adjugate / determinant), then its Cholesky factor: CxInv = L L^T, U = L^T (LLT::matrixU)
    const double c00 = Cx[4] * Cx[8] - Cx[5] * Cx[7], c01 = Cx[5] * Cx[6] - Cx[3] * Cx[8], c02 = Cx[3] * Cx[7] - Cx[4] * Cx[6];
    const double id = 1. / (Cx[0] * c00 + Cx[1] * c01 + Cx[2] * c02);
    const double Ci[9] = {c00 * id, (Cx[2] * Cx[7] - Cx[1] * Cx[8]) * id, (Cx[1] * Cx[5] - Cx[2] * Cx[4]) * id,
                          c01 * id, (Cx[0] * Cx[8] - Cx[2] * Cx[6]) * id, (Cx[2] * Cx[3] - Cx[0] * Cx[5]) * id,
                          c02 * id, (Cx[1] * Cx[6] - Cx[0] * Cx[7]) * id, (Cx[0] * Cx[4] - Cx[1] * Cx[3]) * id};
    double L[9] = {0.};
    for (int r = 0; r < 3; r++)
        for (int q = 0; q <= r; q++) {
            double v = Ci[3 * r + q];
            for (int k = 0; k < q; k++) v -= L[3 * r + k] * L[3 * q + k];
            L[3 * r + q] = r == q ? std::sqrt(v) : v / L[3 * q + q];
        }
    for (int k = 0; k < 36; k++) b.A[k] = 0.;
    b.A[0] = L[0];  b.A[1] = L[3];              // U(0,0) U(0,1)
    b.A[7] = L[4];                              // U(1,1)   (U(1,0) = 0)
    b.A[5] = L[6];  b.A[11] = L[7];             // topRightCorner<2,1>() of the 6x6 = column 5: U(0,2), U(1,2)
    b.A[14] = 1. / lambda;                      // _A(2,2)
    b.A[21] = 1. / lambda; b.A[28] = 1. / lambda; b.A[35] = L[8];   // diag(1/lambda, 1/lambda, U(2,2))
}

// constructor, calib_cost_functions.cpp:119-167
inline Block make_block(int tf, int64_t i, double errV, double errW, double lambda, const double *xi1, const double *xi2)
{
    Block b;
    b.tf = tf;
    b.i = i;
    Array6d a, c;
    for (int k = 0; k < 6; k++) { a[k] = xi1[k]; c[k] = xi2[k]; }
    const Array6d z = vgth::inverse_compose(a, c);
    for (int k = 0; k < 6; k++) b.zeta[k] = z[k];
    weight_from_zeta(b, errV, errW, lambda);
    return b;
}

// ---- OdometryCost -------------------------------------------------------------------------------------------
// tf0n_jac_calc, odometry_cost_function.cpp:72-94 (odom_zeta_i :10-36, zeta_i_jacobian :39-69): the poses 0T1 .. 0Tn of
// the wheel increments under intrinsics [r1, r2, g] and d(zeta_i)/d(intrinsics) of every step
inline void wheel_chain(const std::vector<double> &dq, const double *intr, std::vector<Array6d> &tf0, std::vector<std::array<double, 9>> &jz)
{
    const double r1 = intr[0], r2 = intr[1], g = intr[2];
    const size_t n = dq.size() / 2;
    tf0.resize(n);
    jz.resize(n);
    Array6d acc = {0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        const double dl = dq[2 * i], dr = dq[2 * i + 1];
        const double v = (r1 / 2) * dl + (r2 / 2) * dr;
        const double w = (-(r1 / g)) * dl + (r2 / g) * dr;
        acc = vgth::compose(acc, Array6d{v, 0., 0., 0., 0., w});
        tf0[i] = acc;
        jz[i] = {dl / 2, dr / 2, 0, 0, 0, 0, -dl / g, dr / g, (r1 * dl - r2 * dr) / (g * g)};
    }
}

// constructor, odometry_cost_function.cpp:147-197
inline Block make_cost_block(int tf, int64_t i, double errV, double errW, double lambda, const double *dq, int n, const double *intr_prior,
                             int pblock)
{
    Block b;
    b.tf = tf;
    b.i = i;
    b.pblock = pblock;
    b.dq.assign(dq, dq + 2 * (size_t)n);
    std::vector<Array6d> tf0;
    std::vector<std::array<double, 9>> jz;
    wheel_chain(b.dq, intr_prior, tf0, jz);
    for (int k = 0; k < 6; k++) b.zeta[k] = tf0.back()[k];
    weight_from_zeta(b, errV, errW, lambda);
    return b;
}

// Evaluate, calib_cost_functions.cpp:171-212.  J1 / J2 row-major 6x6, may be NULL.
inline void evaluate(const Block &b, const double *xi1, const double *xi2, double *res, double *J1, double *J2)
{
    Array6d a, c, zp;
    for (int k = 0; k < 6; k++) { a[k] = xi1[k]; c[k] = xi2[k]; zp[k] = b.zeta[k]; }
    const Array6d zeta = vgth::inverse_compose(a, c);
    const Array6d err = vgth::inverse_compose(zp, zeta);
    for (int r = 0; r < 6; r++) {
        double s = 0.;
        for (int k = 0; k < 6; k++) s += b.A[6 * r + k] * err[k];
        res[r] = s;
    }
    auto rinv_and_rm = [](const double *xi, double *R, double *RM) {
        const vg::RotTrig g = vg::rot_trig(xi + 3, true, true);
        double M[9];
        vg::rotation_matrix(xi + 3, -1., g, R);   // rotMatInv
        vg::inter_omega_rot(xi + 3, g, M);
        vg::mat3_mul(R, M, RM);
    };
    auto blockdiag = [](const double *Ra, const double *Rb, double *out) {
        for (int k = 0; k < 36; k++) out[k] = 0.;
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) {
                out[6 * r + q] = Ra[3 * r + q];
                out[6 * (3 + r) + 3 + q] = Rb[3 * r + q];
            }
    };
    if (J1) {
        double R10[9], RM[9], Jm[36], Rz[9], RH[9], TT[36], T1[36], T2[36];
        rinv_and_rm(xi1, R10, RM);
        blockdiag(R10, RM, Jm);
        const vg::RotTrig gz = vg::rot_trig(zeta.data() + 3, true, false);
        vg::rotation_matrix(zeta.data() + 3, -1., gz, Rz);
        const double H[9] = {0, -zeta[2], zeta[1], zeta[2], 0, -zeta[0], -zeta[1], zeta[0], 0};   // hat(zeta.trans)
        vg::mat3_mul(Rz, H, RH);
        for (int k = 0; k < 36; k++) TT[k] = 0.;                                                  // screwTransfInv
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) {
                TT[6 * r + q] = Rz[3 * r + q];
                TT[6 * r + 3 + q] = -RH[3 * r + q];
                TT[6 * (3 + r) + 3 + q] = Rz[3 * r + q];
            }
        mat6_mul(b.A, TT, T1);
        mat6_mul(T1, Jm, T2);
        for (int k = 0; k < 36; k++) J1[k] = -T2[k];
    }
    if (J2) {
        double R20[9], RM[9], Jm[36];
        rinv_and_rm(xi2, R20, RM);
        blockdiag(R20, RM, Jm);
        mat6_mul(b.A, Jm, J2);
    }
}

// Evaluate, odometry_cost_function.cpp:202-266 (parameter blocks xi1[6], xi2[6], intrinsics[3]); J3 row-major 6 x 3
inline void evaluate_cost(const Block &b, const double *xi1, const double *xi2, const double *intr, double *res, double *J1, double *J2,
                          double *J3)
{
    Array6d a, c;
    for (int k = 0; k < 6; k++) { a[k] = xi1[k]; c[k] = xi2[k]; }
    const Array6d zeta = vgth::inverse_compose(a, c);
    std::vector<Array6d> tf0;
    std::vector<std::array<double, 9>> jz;
    wheel_chain(b.dq, intr, tf0, jz);
    const Array6d zeta_odo = tf0.back();
    const Array6d delta = vgth::inverse_compose(zeta_odo, zeta);
    for (int r = 0; r < 6; r++) {
        double s = 0.;
        for (int k = 0; k < 6; k++) s += b.A[6 * r + k] * delta[k];
        res[r] = s;
    }
    if (J1 || J2) {  // the two pose blocks are OdometryPrior's (:231-252 == calib_cost_functions.cpp:187-209)
        double r_[6];
        Block tmp = b;
        for (int k = 0; k < 6; k++) tmp.zeta[k] = 0.;
        evaluate(tmp, xi1, xi2, r_, J1, J2);
    }
    if (J3) {
        // calc_acc :96-144
        double ACC[9] = {0.};
        const size_t n = tf0.size();
        for (size_t i = 0; i < n; i++) {
            const Array6d tf0j = i > 0 ? tf0[i - 1] : Array6d{0, 0, 0, 0, 0, 0};
            double R0j[9], T[9], T2[9];
            const vg::RotTrig g0 = vg::rot_trig(tf0j.data() + 3, true, false);
            vg::rotation_matrix(tf0j.data() + 3, 1., g0, R0j);
            const Array6d tin = vgth::compose(vgth::inverse(tf0[i]), zeta_odo);
            const double Jm[9] = {1, 0, -tin[1], 0, 1, tin[0], 0, 0, 1};
            vg::mat3_mul(R0j, Jm, T);
            vg::mat3_mul(T, jz[i].data(), T2);
            for (int k = 0; k < 9; k++) ACC[k] = ACC[k] + T2[k];
        }
        const double acc63[18] = {ACC[0], ACC[1], ACC[2], ACC[3], ACC[4], ACC[5], 0, 0, 0, 0, 0, 0, 0, 0, 0, ACC[6], ACC[7], ACC[8]};
        double R31[9], M[9], RM[9], J3m[36], Rd[9], RH[9], TT[36], T1[36], T2m[36];
        const vg::RotTrig gz = vg::rot_trig(zeta_odo.data() + 3, true, true);
        vg::rotation_matrix(zeta_odo.data() + 3, -1., gz, R31);
        vg::inter_omega_rot(zeta_odo.data() + 3, gz, M);
        vg::mat3_mul(R31, M, RM);
        for (int k = 0; k < 36; k++) J3m[k] = 0.;
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) {
                J3m[6 * r + q] = R31[3 * r + q];
                J3m[6 * (3 + r) + 3 + q] = RM[3 * r + q];
            }
        const vg::RotTrig gd = vg::rot_trig(delta.data() + 3, true, false);
        vg::rotation_matrix(delta.data() + 3, -1., gd, Rd);
        const double H[9] = {0, -delta[2], delta[1], delta[2], 0, -delta[0], -delta[1], delta[0], 0};
        vg::mat3_mul(Rd, H, RH);
        for (int k = 0; k < 36; k++) TT[k] = 0.;
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) {
                TT[6 * r + q] = Rd[3 * r + q];
                TT[6 * r + 3 + q] = -RH[3 * r + q];
                TT[6 * (3 + r) + 3 + q] = Rd[3 * r + q];
            }
        mat6_mul(b.A, TT, T1);
        mat6_mul(T1, J3m, T2m);
        for (int r = 0; r < 6; r++)
            for (int q = 0; q < 3; q++) {
                double s = 0.;
                for (int k = 0; k < 6; k++) s += T2m[6 * r + k] * acc63[3 * k + q];
                J3[3 * r + q] = -s;
            }
    }
}

}  // namespace vgodo
