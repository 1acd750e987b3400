// vg_solver_coupled.hpp -- the host side of what is NOT block diagonal in the pose part of the normal equations: the
// TransformationPrior residual and the odometry-coupled sequences (OdometryPrior / OdometryCost tie element i of a sequence to
// i + 1: block tridiagonal elimination on the host, rows handed back to the device path).  Part of the solver translation
// unit (vg_solver_tu.hip); see vg_solver_impl.hpp for the driver.
#pragma once

#include <cmath>
#include <cstring>
#include <vector>

#include "vg_internal.hpp"
#include "vg_transf_host.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// A sequence transform whose consecutive elements are coupled by OdometryPrior blocks: its pose system is block
// tridiagonal (6x6 blocks) instead of block diagonal.  The GPU hands over the raw V_i, g_i, W_i^T of every element;
// the host adds the odometry terms, factors  H = L L^T  (block bidiagonal L), forms the rows  Y = L^-1 [W^T | g]  that
// take the place of the per-pose rows in the Schur complement, and later back-substitutes.  O(n * 6^2 * G) per
// iteration on one core: meant for the few-hundred-pose odometry sets the reference targets.
// ------------------------------------------------------------------------------------------
// TransformationPrior::Evaluate, calib_cost_functions.cpp:214-228: r = A * blockdiag(R, R) * (prior^-1 o xi); J = A
inline void prior_residual(const vgi::Prior &pr, const double *xi6, double *r)
{
    vgth::Array6d prior, xi;
    for (int k = 0; k < 6; k++) { prior[k] = pr.xi[k]; xi[k] = xi6[k]; }
    const vgth::Array6d e = vgth::inverse_compose(prior, xi);
    double er[6];
    for (int k = 0; k < 3; k++) {
        er[k] = pr.R[3 * k] * e[0] + pr.R[3 * k + 1] * e[1] + pr.R[3 * k + 2] * e[2];
        er[3 + k] = pr.R[3 * k] * e[3] + pr.R[3 * k + 1] * e[4] + pr.R[3 * k + 2] * e[5];
    }
    for (int k = 0; k < 6; k++) {
        r[k] = 0.;
        for (int c = 0; c < 6; c++) r[k] += pr.A[6 * k + c] * er[c];
    }
}

struct CoupledSeq {
    int tf = -1;
    int64_t pb = 0, n = 0, param_off = 0;   // first pose block, number of elements, first parameter of the range
    std::vector<vgodo::Block> blocks;       // sorted by element index; OdometryCost blocks carry pblock >= 0
    std::vector<int> pb_goff;               // global column of every parameter block (shared by all sequences)
    std::vector<std::pair<int64_t, vgi::Prior>> unary;  // TransformationPrior blocks on single elements of the range
    std::vector<unsigned char> frozen;      // per element
    std::vector<double> x, xc;              // current / candidate values [n][6]
    std::vector<double> Cf, Bs;             // per element: Cholesky factor C_i (lower, 6x6 row-major), B_i = L_{i+1,i}
    std::vector<double> Y, g, Dd;           // rows [6n][C], full gradient [6n], clamped undamped diagonal [6n]

    // xg: values of the global columns at the same point (the odometry intrinsics of OdometryCost blocks live there)
    double cost2(const std::vector<double> &xv, const double *xg) const
    {
        double c = 0.;
        for (const auto &b : blocks) {
            double r[6];
            if (b.pblock >= 0)
                vgodo::evaluate_cost(b, &xv[(size_t)b.i * 6], &xv[(size_t)(b.i + 1) * 6], xg + pb_goff[(size_t)b.pblock], r, nullptr, nullptr, nullptr);
            else
                vgodo::evaluate(b, &xv[(size_t)b.i * 6], &xv[(size_t)(b.i + 1) * 6], r, nullptr, nullptr);
            for (int k = 0; k < 6; k++) c += r[k] * r[k];
        }
        for (const auto &u : unary) {
            double r[6];
            prior_residual(u.second, &xv[(size_t)u.first * 6], r);
            for (int k = 0; k < 6; k++) c += r[k] * r[k];
        }
        return c;
    }

    static bool chol6(const double *A, double *L)
    {
        for (int k = 0; k < 36; k++) L[k] = 0.;
        for (int r = 0; r < 6; r++)
            for (int c = 0; c <= r; c++) {
                double s = A[6 * r + c];
                for (int k = 0; k < c; k++) s -= L[6 * r + k] * L[6 * c + k];
                if (r == c) {
                    if (!(s > 0.) || !std::isfinite(s)) return false;
                    L[6 * r + r] = std::sqrt(s);
                } else {
                    L[6 * r + c] = s / L[6 * c + c];
                }
            }
        return true;
    }

    // rec: [n][kPoseRec] raw V (packed lower) | g | diag ;  raw: [6n][C] raw W^T | g columns.  Returns false when a
    // diagonal block is not positive definite.
    // OdometryCost blocks: what they add to the GLOBAL part of the normal equations (J3^T J3, J3^T r); their pose and
    // pose-global parts are handled by eliminate()
    void add_global_terms(const std::vector<double> &xv, const double *xg, int G, std::vector<double> &Uo, std::vector<double> &go) const
    {
        for (const auto &b : blocks) {
            if (b.pblock < 0) continue;
            const int g0 = pb_goff[(size_t)b.pblock];
            double r[6], J3[18];
            vgodo::evaluate_cost(b, &xv[(size_t)b.i * 6], &xv[(size_t)(b.i + 1) * 6], xg + g0, r, nullptr, nullptr, J3);
            for (int a2 = 0; a2 < 3; a2++) {
                for (int b2 = 0; b2 < 3; b2++) {
                    double h = 0.;
                    for (int k = 0; k < 6; k++) h += J3[3 * k + a2] * J3[3 * k + b2];
                    Uo[(size_t)(g0 + a2) * G + g0 + b2] += h;
                }
                double gs = 0.;
                for (int k = 0; k < 6; k++) gs += J3[3 * k + a2] * r[k];
                go[(size_t)(g0 + a2)] += gs;
            }
        }
    }

    bool eliminate(const double *rec, const double *raw, int G, double mu, double dmin, double dmax, const double *xg)
    {
        const int C = G + 1;
        std::vector<double> Wodo;  // [6n][G] pose-global coupling of the OdometryCost blocks (J_pose^T J3), if any
        std::vector<double> H((size_t)n * 36, 0.), E((size_t)n * 36, 0.);  // diagonal blocks, E_i = H_{i,i+1}
        g.assign((size_t)n * 6, 0.);
        for (int64_t i = 0; i < n; i++) {
            const double *r = rec + (size_t)i * vg::kPoseRec;
            for (int a2 = 0; a2 < 6; a2++)
                for (int b2 = 0; b2 <= a2; b2++) H[(size_t)i * 36 + 6 * a2 + b2] = H[(size_t)i * 36 + 6 * b2 + a2] = r[a2 * (a2 + 1) / 2 + b2];
            for (int k = 0; k < 6; k++) g[(size_t)i * 6 + k] = r[21 + k];
        }
        for (const auto &b : blocks) {
            double r[6], J1[36], J2[36];
            if (b.pblock >= 0) {
                double J3[18];
                const int g0 = pb_goff[(size_t)b.pblock];
                vgodo::evaluate_cost(b, &x[(size_t)b.i * 6], &x[(size_t)(b.i + 1) * 6], xg + g0, r, J1, J2, J3);
                if (Wodo.empty()) Wodo.assign((size_t)n * 6 * G, 0.);
                for (int a2 = 0; a2 < 6; a2++)
                    for (int c3 = 0; c3 < 3; c3++) {
                        double s1 = 0., s2 = 0.;
                        for (int k = 0; k < 6; k++) {
                            s1 += J1[6 * k + a2] * J3[3 * k + c3];
                            s2 += J2[6 * k + a2] * J3[3 * k + c3];
                        }
                        Wodo[((size_t)b.i * 6 + a2) * G + g0 + c3] += s1;
                        Wodo[((size_t)(b.i + 1) * 6 + a2) * G + g0 + c3] += s2;
                    }
            } else {
                vgodo::evaluate(b, &x[(size_t)b.i * 6], &x[(size_t)(b.i + 1) * 6], r, J1, J2);
            }
            double *H1 = &H[(size_t)b.i * 36], *H2 = &H[(size_t)(b.i + 1) * 36], *Ei = &E[(size_t)b.i * 36];
            for (int a2 = 0; a2 < 6; a2++) {
                for (int b2 = 0; b2 < 6; b2++) {
                    double s11 = 0., s22 = 0., s12 = 0.;
                    for (int k = 0; k < 6; k++) {
                        s11 += J1[6 * k + a2] * J1[6 * k + b2];
                        s22 += J2[6 * k + a2] * J2[6 * k + b2];
                        s12 += J1[6 * k + a2] * J2[6 * k + b2];
                    }
                    H1[6 * a2 + b2] += s11;
                    H2[6 * a2 + b2] += s22;
                    Ei[6 * a2 + b2] += s12;
                }
                double g1 = 0., g2 = 0.;
                for (int k = 0; k < 6; k++) { g1 += J1[6 * k + a2] * r[k]; g2 += J2[6 * k + a2] * r[k]; }
                g[(size_t)b.i * 6 + a2] += g1;
                g[(size_t)(b.i + 1) * 6 + a2] += g2;
            }
        }
        for (const auto &u : unary) {  // constant Jacobian A
            double r[6];
            prior_residual(u.second, &x[(size_t)u.first * 6], r);
            const double *A = u.second.A;
            for (int a2 = 0; a2 < 6; a2++) {
                for (int b2 = 0; b2 < 6; b2++) {
                    double s2 = 0.;
                    for (int k = 0; k < 6; k++) s2 += A[6 * k + a2] * A[6 * k + b2];
                    H[(size_t)u.first * 36 + 6 * a2 + b2] += s2;
                }
                double g1 = 0.;
                for (int k = 0; k < 6; k++) g1 += A[6 * k + a2] * r[k];
                g[(size_t)u.first * 6 + a2] += g1;
            }
        }
        Dd.assign((size_t)n * 6, 0.);
        std::vector<double> R((size_t)n * 6 * C);  // right-hand sides [W^T | g] with the odometry gradient in the last column
        for (int64_t i = 0; i < n; i++) {
            for (int k = 0; k < 6; k++) {
                for (int c = 0; c < G; c++)
                    R[((size_t)i * 6 + k) * C + c] = raw[((size_t)i * 6 + k) * C + c] + (Wodo.empty() ? 0. : Wodo[((size_t)i * 6 + k) * G + c]);
                R[((size_t)i * 6 + k) * C + G] = g[(size_t)i * 6 + k];
            }
            if (frozen[(size_t)i]) {  // constant element: unit block, no coupling, zero right-hand side
                for (int k = 0; k < 36; k++) H[(size_t)i * 36 + k] = 0.;
                for (int k = 0; k < 6; k++) H[(size_t)i * 36 + 7 * k] = 1.;
                for (int k = 0; k < 36; k++) E[(size_t)i * 36 + k] = 0.;
                if (i > 0) for (int k = 0; k < 36; k++) E[(size_t)(i - 1) * 36 + k] = 0.;
                for (int k = 0; k < 6 * C; k++) R[(size_t)i * 6 * C + k] = 0.;
                for (int k = 0; k < 6; k++) g[(size_t)i * 6 + k] = 0.;
            } else {
                for (int k = 0; k < 6; k++) {
                    const double d = H[(size_t)i * 36 + 7 * k];
                    const double dc = d < dmin ? dmin : (d > dmax ? dmax : d);
                    Dd[(size_t)i * 6 + k] = dc;
                    H[(size_t)i * 36 + 7 * k] += mu * dc;
                }
            }
        }
        Cf.assign((size_t)n * 36, 0.);
        Bs.assign((size_t)n * 36, 0.);
        Y.assign((size_t)n * 6 * C, 0.);
        for (int64_t i = 0; i < n; i++) {
            double D[36];
            for (int k = 0; k < 36; k++) D[k] = H[(size_t)i * 36 + k];
            if (i > 0) {  // D -= B_{i-1} B_{i-1}^T
                const double *B = &Bs[(size_t)(i - 1) * 36];
                for (int a2 = 0; a2 < 6; a2++)
                    for (int b2 = 0; b2 < 6; b2++) {
                        double s2 = 0.;
                        for (int k = 0; k < 6; k++) s2 += B[6 * a2 + k] * B[6 * b2 + k];
                        D[6 * a2 + b2] -= s2;
                    }
            }
            double *Ci = &Cf[(size_t)i * 36];
            if (!chol6(D, Ci)) return false;
            // Y_i = C_i^-1 (R_i - B_{i-1} Y_{i-1})
            for (int c = 0; c < C; c++) {
                double v[6];
                for (int k = 0; k < 6; k++) {
                    double s2 = R[((size_t)i * 6 + k) * C + c];
                    if (i > 0)
                        for (int q = 0; q < 6; q++) s2 -= Bs[(size_t)(i - 1) * 36 + 6 * k + q] * Y[((size_t)(i - 1) * 6 + q) * C + c];
                    v[k] = s2;
                }
                for (int k = 0; k < 6; k++) {
                    double s2 = v[k];
                    for (int q = 0; q < k; q++) s2 -= Ci[6 * k + q] * Y[((size_t)i * 6 + q) * C + c];
                    Y[((size_t)i * 6 + k) * C + c] = s2 / Ci[6 * k + k];
                }
            }
            if (i + 1 < n) {  // B_i = E_i^T C_i^-T  <=>  B_i C_i^T = E_i^T : row by row forward substitution
                double *B = &Bs[(size_t)i * 36];
                const double *Ei = &E[(size_t)i * 36];
                for (int a2 = 0; a2 < 6; a2++)
                    for (int k = 0; k < 6; k++) {
                        double s2 = Ei[6 * k + a2];  // (E^T)[a2][k]
                        for (int q = 0; q < k; q++) s2 -= B[6 * a2 + q] * Ci[6 * k + q];
                        B[6 * a2 + k] = s2 / Ci[6 * k + k];
                    }
            }
        }
        return true;
    }

    // dp = -L^-T (y + Y dg); returns the scalar terms gp.dp | sum D dp^2 | |dp|^2 | |g|^2 | max|g|
    void backsub(const double *dg, int G, std::vector<double> &dp, double *scal5) const
    {
        const int C = G + 1;
        dp.assign((size_t)n * 6, 0.);
        std::vector<double> v((size_t)n * 6);
        for (int64_t i = 0; i < n; i++)
            for (int k = 0; k < 6; k++) {
                double s2 = Y[((size_t)i * 6 + k) * C + G];
                for (int c = 0; c < G; c++) s2 += Y[((size_t)i * 6 + k) * C + c] * dg[c];
                v[(size_t)i * 6 + k] = s2;
            }
        std::vector<double> xs((size_t)n * 6, 0.);
        for (int64_t i = n - 1; i >= 0; i--) {  // L^T x = v :  C_i^T x_i = v_i - B_i^T x_{i+1}
            double w[6];
            for (int k = 0; k < 6; k++) {
                double s2 = v[(size_t)i * 6 + k];
                if (i + 1 < n)
                    for (int q = 0; q < 6; q++) s2 -= Bs[(size_t)i * 36 + 6 * q + k] * xs[(size_t)(i + 1) * 6 + q];
                w[k] = s2;
            }
            const double *Ci = &Cf[(size_t)i * 36];
            for (int k = 5; k >= 0; k--) {
                double s2 = w[k];
                for (int q = k + 1; q < 6; q++) s2 -= Ci[6 * q + k] * xs[(size_t)i * 6 + q];
                xs[(size_t)i * 6 + k] = s2 / Ci[6 * k + k];
            }
        }
        for (int k = 0; k < 5; k++) scal5[k] = 0.;
        for (int64_t i = 0; i < n; i++) {
            if (frozen[(size_t)i]) continue;
            for (int k = 0; k < 6; k++) {
                const double d = -xs[(size_t)i * 6 + k], gk = g[(size_t)i * 6 + k];
                dp[(size_t)i * 6 + k] = d;
                scal5[0] += gk * d;
                scal5[1] += Dd[(size_t)i * 6 + k] * d * d;
                scal5[2] += d * d;
                scal5[3] += gk * gk;
                scal5[4] = std::fabs(gk) > scal5[4] ? std::fabs(gk) : scal5[4];
            }
        }
    }
};

}  // namespace
