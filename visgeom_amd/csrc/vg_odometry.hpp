// vg_odometry.hpp -- host side of the OdometryPrior residual block of the calibration problem
// (struct OdometryPrior, include/calibration/calib_cost_functions.h:64-77; constructor
// src/calibration/calib_cost_functions.cpp:119-167, Evaluate :171-212).  Six residuals between two CONSECUTIVE
// elements of a sequence transform:  r = A * (zetaPrior^-1 o (xi1^-1 o xi2)),  with the reference's Jacobians
//     dr/dxi1 = -A * screwTransfInv(zeta) * blockdiag(R10, R10 M(r1)),   dr/dxi2 = A * blockdiag(R20, R20 M(r2)).
// "few rows, CPU is fine" -- these blocks are evaluated on the host; what they change is
// the structure of the pose system (block tridiagonal inside the sequence), handled in vg_solver_impl.hpp.
#pragma once

#include <cmath>

#include "vg_transf_host.hpp"

namespace vgodo {

using vgth::Array6d;

struct Block {
    int tf = -1;        // sequence transform
    int64_t i = 0;      // couples elements i and i + 1
    double zeta[6];     // xi1_odom^-1 o xi2_odom
    double A[36];       // row-major
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

}  // namespace vgodo
