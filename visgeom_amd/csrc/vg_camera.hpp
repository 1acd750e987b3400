// vg_camera.hpp -- EUCM / UCM / Mei: projection, d(u,v)/dX and d(u,v)/d(intrinsics) of ONE point,
// fused into a single evaluation per corner.
//
// The reference evaluates three virtual functions per corner (projectPoint, projectionJacobian,
// intrinsicJacobian; include/projection/generic_camera.h:39-51), each recomputing rho / eta from
// scratch.  Here one lane computes all three from shared sub-expressions.  Sharing is only done
// where the reference's three copies are the same expression (so the value is bit-identical);
// where they differ in association (UCM/Mei: rho summed z^2+x^2+y^2 in the projector but
// x^2+y^2+z^2 in the Jacobians, ucm.h:44 vs :125) both forms are kept.
#pragma once

#include "vg_geometry.hpp"

namespace vg {

enum Model : int { kEUCM = 0, kUCM = 1, kMEI = 2 };

template <int MODEL>
struct CameraTraits;
template <>
struct CameraTraits<kEUCM> {
    static constexpr int K = 6;
};
template <>
struct CameraTraits<kUCM> {
    static constexpr int K = 5;
};
template <>
struct CameraTraits<kMEI> {
    static constexpr int K = 10;
};

VG_HD int num_intrinsics(int model) { return model == kEUCM ? 6 : model == kUCM ? 5 : model == kMEI ? 10 : -1; }

// Result of one corner.  P = [du/dX (3) | dv/dX (3)], Ju / Jv = the two intrinsic-Jacobian rows.
template <int K>
struct CornerEval {
    double u, v;
    bool ok;  // projectPoint's return value
    double P[6];
    double Ju[K], Jv[K];
};

// ------------------------------------------------------------------------------------------ EUCM
// EnhancedProjector eucm.h:29-63 ; projectionJacobian :115-167 ; intrinsicJacobian :169-226
template <bool WANT_P, bool WANT_I>
VG_HD void eval_corner(const double *__restrict__ p, double x, double y, double z, CornerEval<6> &e,
                       std::integral_constant<int, kEUCM>)
{
    const double alpha = p[0], beta = p[1], fu = p[2], fv = p[3], u0 = p[4], v0 = p[5];
    const double x2y2 = x * x + y * y;
    const double rho = sqrt(z * z + beta * (x2y2));
    const double gamma = 1. - alpha;
    const double eta = alpha * rho + gamma * z;  // == projector's denom

    bool ok = !(eta < 1e-3);
    if (alpha > 0.5) {
        // upper-hemisphere test, eucm.h:43-48
        const double zn = z / eta;
        const double C = (alpha - 1.) / (alpha + alpha - 1.);
        if (zn < C) ok = false;
    }
    e.ok = ok;
    const double xn = x / eta;
    const double yn = y / eta;
    e.u = fu * xn + u0;
    e.v = fv * yn + v0;

    if (WANT_P) {
        const double k = 1. / eta / eta;
        const double abrho = alpha * beta / rho;
        const double Jxy = k * abrho * x * y;
        const double Jz = k * (gamma + alpha * z / rho);
        const double Jx = gamma * z + alpha * rho;
        e.P[0] = ok ? fu * k * (Jx - abrho * x * x) : 0.;
        e.P[1] = ok ? -fu * Jxy : 0.;
        e.P[2] = ok ? -fu * x * Jz : 0.;
        e.P[3] = ok ? -fv * Jxy : 0.;
        e.P[4] = ok ? fv * k * (Jx - abrho * y * y) : 0.;
        e.P[5] = ok ? -fv * y * Jz : 0.;
    }
    if (WANT_I) {
        const double eta2 = eta * eta;
        e.Ju[0] = ok ? -fu * x * (rho - z) / eta2 : 0.;
        e.Ju[1] = ok ? -fu * x * alpha * x2y2 / (2 * eta2 * rho) : 0.;
        e.Ju[2] = ok ? xn : 0.;
        e.Ju[3] = 0.;
        e.Ju[4] = ok ? 1. : 0.;
        e.Ju[5] = 0.;
        e.Jv[0] = ok ? -fv * y * (rho - z) / eta2 : 0.;
        e.Jv[1] = ok ? -fv * y * alpha * x2y2 / (2 * eta2 * rho) : 0.;
        e.Jv[2] = 0.;
        e.Jv[3] = ok ? yn : 0.;
        e.Jv[4] = 0.;
        e.Jv[5] = ok ? 1. : 0.;
    }
}

// normalized-point Jacobian dm/dX shared by ucm.h:120-142 and mei.h:136-156
struct UnifiedPoint {
    double xn, yn, rho, deninv;
    double dm[6];
};

VG_HD UnifiedPoint unified_point(double xi, double x, double y, double z)
{
    UnifiedPoint q;
    const double xx = x * x;
    const double yy = y * y;
    const double zz = z * z;
    q.rho = sqrt(xx + yy + zz);
    const double rhoinv = 1. / q.rho;
    q.deninv = 1. / (xi * q.rho + z);
    const double deninv2 = q.deninv * q.deninv;
    q.xn = x * q.deninv;
    q.yn = y * q.deninv;
    q.dm[0] = (xi * q.rho + z - xi * xx * rhoinv) * deninv2;
    q.dm[1] = -xi * x * y * rhoinv * deninv2;
    q.dm[2] = -x * (1 + xi * z * rhoinv) * deninv2;
    q.dm[3] = -xi * x * y * rhoinv * deninv2;
    q.dm[4] = (xi * q.rho + z - xi * yy * rhoinv) * deninv2;
    q.dm[5] = -y * (1 + xi * z * rhoinv) * deninv2;
    return q;
}

// ------------------------------------------------------------------------------------------ UCM
// UnifiedProjector ucm.h:32-59 ; projectionJacobian :112-151 ; intrinsicJacobian :153-197.
// Never reports failure.
template <bool WANT_P, bool WANT_I>
VG_HD void eval_corner(const double *__restrict__ p, double x, double y, double z, CornerEval<5> &e,
                       std::integral_constant<int, kUCM>)
{
    const double xi = p[0], fu = p[1], fv = p[2], u0 = p[3], v0 = p[4];
    {
        const double rho = sqrt(z * z + x * x + y * y);  // projector's association
        const double denominv = 1. / (z + xi * rho);
        const double xn = x * denominv;
        const double yn = y * denominv;
        e.u = fu * xn + u0;
        e.v = fv * yn + v0;
        e.ok = true;
    }
    if (WANT_P || WANT_I) {
        const UnifiedPoint q = unified_point(xi, x, y, z);
        if (WANT_P) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                e.P[j] = fu * q.dm[j];
                e.P[3 + j] = fv * q.dm[3 + j];
            }
        }
        if (WANT_I) {
            e.Ju[0] = -fu * q.xn * q.deninv * q.rho;
            e.Ju[1] = q.xn;
            e.Ju[2] = 0.;
            e.Ju[3] = 1.;
            e.Ju[4] = 0.;
            e.Jv[0] = -fv * q.yn * q.deninv * q.rho;
            e.Jv[1] = 0.;
            e.Jv[2] = q.yn;
            e.Jv[3] = 0.;
            e.Jv[4] = 1.;
        }
    }
}

// ------------------------------------------------------------------------------------------ Mei
// MeiProjector mei.h:29-68 ; projectionJacobian :121-191 ; intrinsicJacobian :193-285.
template <bool WANT_P, bool WANT_I>
VG_HD void eval_corner(const double *__restrict__ p, double x, double y, double z, CornerEval<10> &e,
                       std::integral_constant<int, kMEI>)
{
    const double xi = p[0], k1 = p[1], k2 = p[2], k3 = p[3], k4 = p[4], k5 = p[5];
    const double fu = p[6], fv = p[7], u0 = p[8], v0 = p[9];
    {
        const double rho = sqrt(z * z + x * x + y * y);
        const double denominv = 1. / (z + xi * rho);
        const double xn = x * denominv;
        const double yn = y * denominv;
        const double xx = xn * xn, xy = xn * yn, yy = yn * yn;
        const double r2 = xx + yy;
        const double D = 1. + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
        const double deltax = 2. * k4 * xy + k5 * (r2 + 2. * xx);
        const double deltay = 2. * k5 * xy + k4 * (r2 + 2. * yy);
        e.u = fu * (xn * D + deltax) + u0;
        e.v = fv * (yn * D + deltay) + v0;
        e.ok = true;
    }
    if (WANT_P || WANT_I) {
        const UnifiedPoint q = unified_point(xi, x, y, z);
        const double xn = q.xn, yn = q.yn;
        const double xxn = xn * xn;
        const double yyn = yn * yn;
        const double xyn = xn * yn;
        const double r2 = yn * yn + xn * xn;
        const double D = 1. + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
        const double dDdr2 = k1 + 2 * k2 * r2 + 3 * k3 * r2 * r2;
        double a0 = D + 2 * xxn * dDdr2 + 2 * k4 * yn + 6 * k5 * xn;
        double a1 = 2 * xyn * dDdr2 + 2 * k4 * xn + 2 * k5 * yn;
        double b0 = 2 * xyn * dDdr2 + 2 * k5 * yn + 2 * k4 * xn;
        double b1 = D + 2 * yyn * dDdr2 + 2 * k5 * xn + 6 * k4 * yn;
        a0 *= fu; a1 *= fu;
        b0 *= fv; b1 *= fv;
        if (WANT_P) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                e.P[j] = a0 * q.dm[j] + a1 * q.dm[3 + j];
                e.P[3 + j] = b0 * q.dm[j] + b1 * q.dm[3 + j];
            }
        }
        if (WANT_I) {
            const double deltax = 2. * k4 * xyn + k5 * (r2 + 2. * xxn);
            const double deltay = 2. * k5 * xyn + k4 * (r2 + 2. * yyn);
            const double xd = xn * D + deltax;
            const double yd = yn * D + deltay;
            const double dxndxi = -xn * q.deninv * q.rho;
            const double dyndxi = -yn * q.deninv * q.rho;
            e.Ju[0] = a0 * dxndxi + a1 * dyndxi;
            e.Ju[1] = fu * xn * r2;
            e.Ju[2] = fu * xn * r2 * r2;
            e.Ju[3] = fu * xn * r2 * r2 * r2;
            e.Ju[4] = 2. * fu * xyn;
            e.Ju[5] = fu * (r2 + 2. * xxn);
            e.Ju[6] = xd;
            e.Ju[7] = 0.;
            e.Ju[8] = 1.;
            e.Ju[9] = 0.;
            e.Jv[0] = b0 * dxndxi + b1 * dyndxi;
            e.Jv[1] = fv * yn * r2;
            e.Jv[2] = fv * yn * r2 * r2;
            e.Jv[3] = fv * yn * r2 * r2 * r2;
            e.Jv[4] = fv * (r2 + 2. * yyn);
            e.Jv[5] = 2. * fv * xyn;
            e.Jv[6] = 0.;
            e.Jv[7] = yd;
            e.Jv[8] = 0.;
            e.Jv[9] = 1.;
        }
    }
}

template <int MODEL, bool WANT_P, bool WANT_I>
VG_HD void eval_corner(const double *__restrict__ p, double x, double y, double z,
                       CornerEval<CameraTraits<MODEL>::K> &e)
{
    eval_corner<WANT_P, WANT_I>(p, x, y, z, e, std::integral_constant<int, MODEL>{});
}

// ------------------------------------------------------------------------------------------
// FAST variants, used only by the compute-bound fused Gram kernel (vg_gram.hpp): the same formulas with FMA
// contraction allowed and, for EUCM, the twelve IEEE divisions replaced by two shared reciprocals (1/eta, 1/rho).
// Results differ from the reference-order evaluation above by a few ulp; the rows never leave the CU and the
// Gram matrices are checked (tests/test_gpu_gram.py) against a long-double Gram of reference-order rows at 1e-10.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void eval_corner_fast(const double *__restrict__ p, double x, double y, double z, CornerEval<6> &e,
                                                 std::integral_constant<int, kEUCM>)
{
#pragma clang fp contract(fast)
    const double alpha = p[0], beta = p[1], fu = p[2], fv = p[3], u0 = p[4], v0 = p[5];
    const double x2y2 = x * x + y * y;
    double rho, ir_raw;
    sqrt_rsqrt_nr(z * z + beta * x2y2, rho, ir_raw);
    const double gamma = 1. - alpha;
    const double eta = alpha * rho + gamma * z;
    const double ie_raw = rcp_nr(eta);
    bool ok = !(eta < 1e-3);
    if (alpha > 0.5) {
        const double C = (alpha - 1.) * rcp_nr(alpha + alpha - 1.);
        if (z * ie_raw < C) ok = false;
    }
    e.ok = ok;
    // failed projection -> zero rows (eucm.h:141-150,198-206).  The two reciprocals are SELECTED to zero, so every
    // masked product below is 0 * finite = exactly 0 (eta == 0 gives an infinite ie_raw that is never used; rho and
    // 1 / rho are finite by construction)
    const double ie = ok ? ie_raw : 0., ir = ok ? ir_raw : 0.;
    const double m = ok ? 1. : 0.;
    const double xn = x * ie, yn = y * ie;
    e.u = fu * xn + u0;
    e.v = fv * yn + v0;
    const double k = ie * ie;
    const double abrho = alpha * beta * ir;
    const double Jxy = k * abrho * x * y;
    const double Jz = k * (gamma + alpha * z * ir);
    const double Jx = gamma * z + alpha * rho;
    const double fuk = fu * k, fvk = fv * k;
    e.P[0] = fuk * (Jx - abrho * x * x);
    e.P[1] = -fu * Jxy;
    e.P[2] = -fu * x * Jz;
    e.P[3] = -fv * Jxy;
    e.P[4] = fvk * (Jx - abrho * y * y);
    e.P[5] = -fv * y * Jz;
    const double db = 0.5 * alpha * x2y2 * k * ir;
    e.Ju[0] = -fuk * x * (rho - z);
    e.Ju[1] = -fu * x * db;
    e.Ju[2] = xn;
    e.Ju[3] = 0.;
    e.Ju[4] = m;
    e.Ju[5] = 0.;
    e.Jv[0] = -fvk * y * (rho - z);
    e.Jv[1] = -fv * y * db;
    e.Jv[2] = 0.;
    e.Jv[3] = yn;
    e.Jv[4] = 0.;
    e.Jv[5] = m;
}

__device__ __forceinline__ void eval_corner_fast(const double *__restrict__ p, double x, double y, double z, CornerEval<5> &e,
                                                 std::integral_constant<int, kUCM>)
{
#pragma clang fp contract(fast)
    const double xi = p[0], fu = p[1], fv = p[2], u0 = p[3], v0 = p[4];
    const double xx = x * x, yy = y * y;
    double rho, ri;
    sqrt_rsqrt_nr(xx + yy + z * z, rho, ri);
    const double di = rcp_nr(xi * rho + z);
    const double d2 = di * di;
    const double xn = x * di, yn = y * di;
    e.ok = true;
    e.u = fu * xn + u0;
    e.v = fv * yn + v0;
    const double den = xi * rho + z, cxy = -xi * x * y * ri * d2, cz = (1. + xi * z * ri) * d2;
    e.P[0] = fu * (den - xi * xx * ri) * d2;
    e.P[1] = fu * cxy;
    e.P[2] = -fu * x * cz;
    e.P[3] = fv * cxy;
    e.P[4] = fv * (den - xi * yy * ri) * d2;
    e.P[5] = -fv * y * cz;
    e.Ju[0] = -fu * xn * di * rho;
    e.Ju[1] = xn;
    e.Ju[2] = 0.;
    e.Ju[3] = 1.;
    e.Ju[4] = 0.;
    e.Jv[0] = -fv * yn * di * rho;
    e.Jv[1] = 0.;
    e.Jv[2] = yn;
    e.Jv[3] = 0.;
    e.Jv[4] = 1.;
}

__device__ __forceinline__ void eval_corner_fast(const double *__restrict__ p, double x, double y, double z, CornerEval<10> &e,
                                                 std::integral_constant<int, kMEI>)
{
#pragma clang fp contract(fast)
    const double xi = p[0], k1 = p[1], k2 = p[2], k3 = p[3], k4 = p[4], k5 = p[5];
    const double fu = p[6], fv = p[7], u0 = p[8], v0 = p[9];
    const double xx = x * x, yy = y * y;
    double rho, ri;
    sqrt_rsqrt_nr(xx + yy + z * z, rho, ri);
    const double di = rcp_nr(xi * rho + z);
    const double d2 = di * di;
    const double xn = x * di, yn = y * di;
    const double xxn = xn * xn, yyn = yn * yn, xyn = xn * yn;
    const double r2 = xxn + yyn;
    const double D = 1. + r2 * (k1 + r2 * (k2 + r2 * k3));
    const double dD = k1 + r2 * (2. * k2 + 3. * k3 * r2);
    const double deltax = 2. * k4 * xyn + k5 * (r2 + 2. * xxn);
    const double deltay = 2. * k5 * xyn + k4 * (r2 + 2. * yyn);
    const double xd = xn * D + deltax, yd = yn * D + deltay;
    e.ok = true;
    e.u = fu * xd + u0;
    e.v = fv * yd + v0;
    const double den = xi * rho + z, cxy = -xi * x * y * ri * d2, cz = (1. + xi * z * ri) * d2;
    const double dm0 = (den - xi * xx * ri) * d2, dm2 = -x * cz, dm4 = (den - xi * yy * ri) * d2, dm5 = -y * cz;
    const double a0 = fu * (D + 2. * xxn * dD + 2. * k4 * yn + 6. * k5 * xn);
    const double a1 = fu * (2. * xyn * dD + 2. * k4 * xn + 2. * k5 * yn);
    const double b0 = fv * (2. * xyn * dD + 2. * k5 * yn + 2. * k4 * xn);
    const double b1 = fv * (D + 2. * yyn * dD + 2. * k5 * xn + 6. * k4 * yn);
    e.P[0] = a0 * dm0 + a1 * cxy;
    e.P[1] = a0 * cxy + a1 * dm4;
    e.P[2] = a0 * dm2 + a1 * dm5;
    e.P[3] = b0 * dm0 + b1 * cxy;
    e.P[4] = b0 * cxy + b1 * dm4;
    e.P[5] = b0 * dm2 + b1 * dm5;
    const double dxi = -di * rho;  // d(xn)/d(xi) = xn * dxi
    const double r4 = r2 * r2;
    e.Ju[0] = (a0 * xn + a1 * yn) * dxi;
    e.Ju[1] = fu * xn * r2;
    e.Ju[2] = fu * xn * r4;
    e.Ju[3] = fu * xn * r4 * r2;
    e.Ju[4] = 2. * fu * xyn;
    e.Ju[5] = fu * (r2 + 2. * xxn);
    e.Ju[6] = xd;
    e.Ju[7] = 0.;
    e.Ju[8] = 1.;
    e.Ju[9] = 0.;
    e.Jv[0] = (b0 * xn + b1 * yn) * dxi;
    e.Jv[1] = fv * yn * r2;
    e.Jv[2] = fv * yn * r4;
    e.Jv[3] = fv * yn * r4 * r2;
    e.Jv[4] = fv * (r2 + 2. * yyn);
    e.Jv[5] = 2. * fv * xyn;
    e.Jv[6] = 0.;
    e.Jv[7] = yd;
    e.Jv[8] = 0.;
    e.Jv[9] = 1.;
}

template <int MODEL>
__device__ __forceinline__ void eval_corner_fast(const double *__restrict__ p, double x, double y, double z,
                            CornerEval<CameraTraits<MODEL>::K> &e)
{
    eval_corner_fast(p, x, y, z, e, std::integral_constant<int, MODEL>{});
}

// pose rows with FMA contraction (same formula as pose_rows)
VG_HD void pose_rows_fast(const double *P, double X0, double X1, double X2, const double *fm, double *out)
{
#pragma clang fp contract(fast)
    const double *R12 = fm, *M12 = fm + 9, *t13 = fm + 18;
    const double a = X0 - t13[0], b = X1 - t13[1], c = X2 - t13[2];
#pragma unroll
    for (int row = 0; row < 2; row++) {
        const double *p = P + 3 * row;
        double *o = out + 6 * row;
#pragma unroll
        for (int j = 0; j < 3; j++) o[j] = p[0] * R12[0 + j] + p[1] * R12[3 + j] + p[2] * R12[6 + j];
        // (-p) * hat(t3X): [-(p1*c - p2*b), -(p2*a - p0*c), -(p0*b - p1*a)]
        const double t0 = p[2] * b - p[1] * c, t1 = p[0] * c - p[2] * a, t2 = p[1] * a - p[0] * b;
#pragma unroll
        for (int j = 0; j < 3; j++) o[3 + j] = t0 * M12[0 + j] + t1 * M12[3 + j] + t2 * M12[6 + j];
    }
}


// InterJacobian::dpdxi  jacobian.h:155-171 : the 2x6 pose-Jacobian rows of chain member `fm`
// (R12[9], M12[9], t13[3]) for a corner with camera-frame point X and projection Jacobian P.
//   out[0..5] = [P0*R12 | ((-P0)*hat(X - t13))*M12],  out[6..11] likewise with P1.
VG_HD void pose_rows(const double *P, double X0, double X1, double X2, const double *fm, double *out)
{
    const double *R12 = fm, *M12 = fm + 9, *t13 = fm + 18;
    const double t3X[3] = {X0 - t13[0], X1 - t13[1], X2 - t13[2]};
    // hat(t3X)  geometry_core.h:126-132
    const double H[9] = {0, -t3X[2], t3X[1], t3X[2], 0, -t3X[0], -t3X[1], t3X[0], 0};
#pragma unroll
    for (int row = 0; row < 2; row++) {
        const double *p = P + 3 * row;
        double *o = out + 6 * row;
#pragma unroll
        for (int j = 0; j < 3; j++) o[j] = p[0] * R12[0 + j] + p[1] * R12[3 + j] + p[2] * R12[6 + j];
        const double n0 = -p[0], n1 = -p[1], n2 = -p[2];
        double tmp[3];
#pragma unroll
        for (int j = 0; j < 3; j++) tmp[j] = n0 * H[0 + j] + n1 * H[3 + j] + n2 * H[6 + j];
#pragma unroll
        for (int j = 0; j < 3; j++) o[3 + j] = tmp[0] * M12[0 + j] + tmp[1] * M12[3 + j] + tmp[2] * M12[6 + j];
    }
}

}  // namespace vg
