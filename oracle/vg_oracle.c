/*
 * vg_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See vg_oracle.h.
 *
 * Plain-C restatement of the reference's arithmetic, expression by expression, in the
 * reference's evaluation order (C/C++ left-to-right for equal precedence; fixed-size matrix
 * products summed k = 0,1,2).  Build with -O2 -ffp-contract=off and NO fast-math so that
 * no expression is re-associated or fused.  All paths below are relative to /root/reference.
 */
#include "vg_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

int vgo_num_intrinsics(int model)
{
    switch (model) {
    case VGO_MODEL_EUCM: return 6;  /* eucm.h:65  INTRINSIC_COUNT */
    case VGO_MODEL_UCM: return 5;   /* ucm.h:61 */
    case VGO_MODEL_MEI: return 10;  /* mei.h:70 */
    default: return -1;
    }
}

/* ------------------------------------------------------------------------------------------
 * geometry
 * ---------------------------------------------------------------------------------------- */

/* Eigen's v.norm() on a 3-vector: sqrt of the sum of squares */
static double norm3(const double v[3])
{
    return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
}

/* include/geometry/geometry_core.h:24-30 */
static double sinc_(double x)
{
    if (x == 0.) return 1.;
    return sin(x) / x;
}

/* include/geometry/geometry_core.h:32-38 */
static double normalize_angle(double th)
{
    if (th > M_PI) return th - 2 * M_PI;
    else if (th < -M_PI) return th + 2 * M_PI;
    else return th;
}

/* include/geometry/quaternion.h:31-50 ; q = (x, y, z, w) */
void vgo_quat_from_rotvec(const double rot[3], double q[4])
{
    double theta = norm3(rot);
    if (fabs(theta) < 1e-6) {
        q[0] = rot[0] / 2.;
        q[1] = rot[1] / 2.;
        q[2] = rot[2] / 2.;
        q[3] = 1.;
    } else {
        double u0 = rot[0] / theta, u1 = rot[1] / theta, u2 = rot[2] / theta;
        double s = sin(theta / 2.);
        q[0] = u0 * s;
        q[1] = u1 * s;
        q[2] = u2 * s;
        q[3] = cos(theta / 2.);
    }
}

/* include/geometry/quaternion.h:84-98 */
void vgo_quat_to_rotvec(const double q[4], double rot[3])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    double s = sqrt(x * x + y * y + z * z);
    if (s < 1e-5) {
        rot[0] = x * 2.;
        rot[1] = y * 2.;
        rot[2] = z * 2.;
    } else {
        double th = 2. * atan2(s, w);
        double thn = normalize_angle(th);
        rot[0] = x / s * thn;
        rot[1] = y / s * thn;
        rot[2] = z / s * thn;
    }
}

/* include/geometry/quaternion.h:61-82 */
static void quat_rotate(const double q[4], const double v[3], double out[3])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    double t1 = w * x;
    double t2 = w * y;
    double t3 = w * z;
    double t4 = -x * x;
    double t5 = x * y;
    double t6 = x * z;
    double t7 = -y * y;
    double t8 = y * z;
    double t9 = -z * z;
    const double v1 = v[0], v2 = v[1], v3 = v[2];
    out[0] = 2. * ((t7 + t9) * v1 + (t5 - t3) * v2 + (t2 + t6) * v3) + v1;
    out[1] = 2. * ((t3 + t5) * v1 + (t4 + t9) * v2 + (t8 - t1) * v3) + v2;
    out[2] = 2. * ((t6 - t2) * v1 + (t1 + t8) * v2 + (t4 + t7) * v3) + v3;
}

/* include/geometry/quaternion.h:105-118 */
static void quat_mul(const double a[4], const double b[4], double out[4])
{
    const double x = a[0], y = a[1], z = a[2], w = a[3];
    const double x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
    double wn = w * w2 - x * x2 - y * y2 - z * z2;
    double xn = w * x2 + x * w2 + y * z2 - z * y2;
    double yn = w * y2 - x * z2 + y * w2 + z * x2;
    double zn = w * z2 + x * y2 - y * x2 + z * w2;
    out[0] = xn;
    out[1] = yn;
    out[2] = zn;
    out[3] = wn;
}

/* include/geometry/transformation.h:80-88 ; a, b, out = [t(3), r(3)] (transformation.h:46) */
void vgo_compose(const double a[6], const double b[6], double out[6])
{
    double q1[4], q2[4], qres[4], rt[3];
    vgo_quat_from_rotvec(a + 3, q1);
    vgo_quat_from_rotvec(b + 3, q2);
    quat_rotate(q1, b, rt);
    double t0 = rt[0] + a[0], t1 = rt[1] + a[1], t2 = rt[2] + a[2];
    quat_mul(q1, q2, qres);
    double r[3];
    vgo_quat_to_rotvec(qres, r);
    out[0] = t0; out[1] = t1; out[2] = t2;
    out[3] = r[0]; out[4] = r[1]; out[5] = r[2];
}

/* include/geometry/transformation.h:101-110 */
void vgo_compose_inverse(const double a[6], const double b[6], double out[6])
{
    double q1[4], q2[4], q2inv[4], qres[4], rt[3];
    vgo_quat_from_rotvec(a + 3, q1);
    vgo_quat_from_rotvec(b + 3, q2);
    q2inv[0] = -q2[0]; q2inv[1] = -q2[1]; q2inv[2] = -q2[2]; q2inv[3] = q2[3]; /* quaternion.h:100-103 */
    quat_mul(q1, q2inv, qres);
    quat_rotate(qres, b, rt);
    double t0 = a[0] - rt[0], t1 = a[1] - rt[1], t2 = a[2] - rt[2];
    double r[3];
    vgo_quat_to_rotvec(qres, r);
    out[0] = t0; out[1] = t1; out[2] = t2;
    out[3] = r[0]; out[4] = r[1]; out[5] = r[2];
}

/* include/geometry/geometry_core.h:40-76 ; R row-major 3x3 */
void vgo_rotation_matrix(const double v[3], double R[9])
{
    double th = norm3(v);
    if (th < 1e-5) {
        R[0] = 1.;    R[1] = -v[2]; R[2] = v[1];
        R[3] = v[2];  R[4] = 1.;    R[5] = -v[0];
        R[6] = -v[1]; R[7] = v[0];  R[8] = 1.;
    } else {
        double thInv = 1. / th;
        double u1 = v[0] * thInv;
        double u2 = v[1] * thInv;
        double u3 = v[2] * thInv;
        double sinth = sin(th);
        double costhVar = 1. - cos(th);

        R[0] = 1. + costhVar * (u1 * u1 - 1.);
        R[4] = 1. + costhVar * (u2 * u2 - 1.);
        R[8] = 1. + costhVar * (u3 * u3 - 1.);

        R[1] = -sinth * u3 + costhVar * u1 * u2;
        R[2] = sinth * u2 + costhVar * u1 * u3;
        R[5] = -sinth * u1 + costhVar * u2 * u3;

        R[3] = sinth * u3 + costhVar * u2 * u1;
        R[6] = -sinth * u2 + costhVar * u3 * u1;
        R[7] = sinth * u1 + costhVar * u3 * u2;
    }
}

/* include/geometry/geometry_core.h:126-132 */
static void hat_(const double u[3], double M[9])
{
    M[0] = 0;     M[1] = -u[2]; M[2] = u[1];
    M[3] = u[2];  M[4] = 0;     M[5] = -u[0];
    M[6] = -u[1]; M[7] = u[0];  M[8] = 0;
}

/* C = A*B, 3x3 row-major, each coefficient summed k = 0,1,2 */
static void mat3_mul(const double A[9], const double B[9], double C[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C[3 * i + j] = A[3 * i + 0] * B[0 + j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

/* include/geometry/geometry_core.h:158-180 */
void vgo_inter_omega_rot(const double v[3], double B[9])
{
    double theta = norm3(v);
    if (theta < 1e-5) {
        double h0 = v[0] / 2., h1 = v[1] / 2., h2 = v[2] / 2.;
        B[0] = 1.;  B[1] = -h2; B[2] = h1;
        B[3] = h2;  B[4] = 1.;  B[5] = -h0;
        B[6] = -h1; B[7] = h0;  B[8] = 1.;
    } else {
        double u[3] = {v[0] / theta, v[1] / theta, v[2] / theta};
        double uhat[9];
        hat_(u, uhat);
        double thetaHalf = theta / 2.;
        double K1 = sinc_(thetaHalf);
        K1 = thetaHalf * K1 * K1;
        double K2 = (1. - sinc_(theta));
        /* B = Identity + K1*uhat + K2*uhat*uhat  ==  (I + K1*uhat) + ((K2*uhat)*uhat) */
        double k2u[9], prod[9];
        for (int i = 0; i < 9; i++) k2u[i] = K2 * uhat[i];
        mat3_mul(k2u, uhat, prod);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double id = (i == j) ? 1. : 0.;
                B[3 * i + j] = (id + K1 * uhat[3 * i + j]) + prod[3 * i + j];
            }
    }
}

/* ------------------------------------------------------------------------------------------
 * cameras
 * ---------------------------------------------------------------------------------------- */

/* EnhancedProjector  include/projection/eucm.h:29-63 */
static int eucm_project(const double *p, const double X[3], double uv[2])
{
    const double alpha = p[0], beta = p[1], fu = p[2], fv = p[3], u0 = p[4], v0 = p[5];
    const double x = X[0], y = X[1], z = X[2];
    double denom = alpha * sqrt(z * z + beta * (x * x + y * y)) + (1. - alpha) * z;
    if (denom < 1e-3) return 0;
    if (alpha > 0.5) {
        const double zn = z / denom;
        const double C = (alpha - 1.) / (alpha + alpha - 1.);
        if (zn < C) return 0;
    }
    const double xn = x / denom;
    const double yn = y / denom;
    uv[0] = fu * xn + u0;
    uv[1] = fv * yn + v0;
    return 1;
}

/* EnhancedCamera::projectionJacobian  include/projection/eucm.h:115-167 */
static int eucm_projection_jacobian(const double *p, const double X[3], double dudx[3], double dvdx[3])
{
    const double alpha = p[0], beta = p[1], fu = p[2], fv = p[3];
    const double x = X[0], y = X[1], z = X[2];
    double rho = sqrt(z * z + beta * (x * x + y * y));
    double gamma = 1. - alpha;
    double eta = alpha * rho + gamma * z;

    int isProjected = 1;
    if (eta < 1e-3) isProjected = 0;
    else if (alpha > 0.5) {
        const double zn = z / eta;
        const double C = (alpha - 1.) / (alpha + alpha - 1.);
        if (zn < C) isProjected = 0;
    }
    if (!isProjected) {
        dudx[0] = 0; dudx[1] = 0; dudx[2] = 0;
        dvdx[0] = 0; dvdx[1] = 0; dvdx[2] = 0;
        return 0;
    }
    double k = 1. / eta / eta;
    double abrho = alpha * beta / rho;
    double Jxy = k * abrho * x * y;
    double Jz = k * (gamma + alpha * z / rho);
    double Jx = gamma * z + alpha * rho;
    dudx[0] = fu * k * (Jx - abrho * x * x);
    dudx[1] = -fu * Jxy;
    dudx[2] = -fu * x * Jz;
    dvdx[0] = -fv * Jxy;
    dvdx[1] = fv * k * (Jx - abrho * y * y);
    dvdx[2] = -fv * y * Jz;
    return 1;
}

/* EnhancedCamera::intrinsicJacobian  include/projection/eucm.h:169-226 */
static int eucm_intrinsic_jacobian(const double *p, const double X[3], double *du, double *dv)
{
    const double alpha = p[0], beta = p[1], fu = p[2], fv = p[3];
    const double x = X[0], y = X[1], z = X[2];
    double x2y2 = x * x + y * y;
    double rho2 = z * z + beta * (x2y2);
    double rho = sqrt(rho2);
    double gamma = 1. - alpha;
    double eta = alpha * rho + gamma * z;

    int isProjected = 1;
    if (eta < 1e-3) isProjected = 0;
    else if (alpha > 0.5) {
        const double zn = z / eta;
        const double C = (alpha - 1.) / (alpha + alpha - 1.);
        if (zn < C) isProjected = 0;
    }
    if (!isProjected) {
        for (int i = 0; i < 6; i++) { du[i] = 0; dv[i] = 0; }
        return 0;
    }
    double eta2 = eta * eta;
    du[0] = -fu * x * (rho - z) / eta2;
    du[1] = -fu * x * alpha * x2y2 / (2 * eta2 * rho);
    du[2] = x / eta;
    du[3] = 0;
    du[4] = 1;
    du[5] = 0;

    dv[0] = -fv * y * (rho - z) / eta2;
    dv[1] = -fv * y * alpha * x2y2 / (2 * eta2 * rho);
    dv[2] = 0;
    dv[3] = y / eta;
    dv[4] = 0;
    dv[5] = 1;
    return 1;
}

/* UnifiedProjector  include/projection/ucm.h:32-59 (never reports failure) */
static int ucm_project(const double *p, const double X[3], double uv[2])
{
    const double xi = p[0], fu = p[1], fv = p[2], u0 = p[3], v0 = p[4];
    const double x = X[0], y = X[1], z = X[2];
    double rho = sqrt(z * z + x * x + y * y);
    double denominv = 1. / (z + xi * rho);
    double xn = x * denominv;
    double yn = y * denominv;
    uv[0] = fu * xn + u0;
    uv[1] = fv * yn + v0;
    return 1;
}

/* the normalized-point Jacobian dm/dX shared verbatim by ucm.h:120-142 and mei.h:136-156 */
static void unified_dm(double xi, const double X[3], double dm[6], double *xn_, double *yn_,
                       double *rho_, double *deninv_)
{
    const double x = X[0], y = X[1], z = X[2];
    double xx = x * x;
    double yy = y * y;
    double zz = z * z;
    double rho = sqrt(xx + yy + zz);
    double rhoinv = 1. / rho;
    double deninv = 1. / (xi * rho + z);
    double deninv2 = deninv * deninv;
    *xn_ = x * deninv;
    *yn_ = y * deninv;
    *rho_ = rho;
    *deninv_ = deninv;
    dm[0] = (xi * rho + z - xi * xx * rhoinv) * deninv2;
    dm[1] = -xi * x * y * rhoinv * deninv2;
    dm[2] = -x * (1 + xi * z * rhoinv) * deninv2;
    dm[3] = -xi * x * y * rhoinv * deninv2;
    dm[4] = (xi * rho + z - xi * yy * rhoinv) * deninv2;
    dm[5] = -y * (1 + xi * z * rhoinv) * deninv2;
}

/* UnifiedCamera::projectionJacobian  include/projection/ucm.h:112-151 */
static int ucm_projection_jacobian(const double *p, const double X[3], double dudx[3], double dvdx[3])
{
    const double xi = p[0], fu = p[1], fv = p[2];
    double dm[6], xn, yn, rho, deninv;
    unified_dm(xi, X, dm, &xn, &yn, &rho, &deninv);
    for (int j = 0; j < 3; j++) {
        dudx[j] = fu * dm[j];
        dvdx[j] = fv * dm[3 + j];
    }
    return 1;
}

/* UnifiedCamera::intrinsicJacobian  include/projection/ucm.h:153-197 */
static int ucm_intrinsic_jacobian(const double *p, const double X[3], double *du, double *dv)
{
    const double xi = p[0], fu = p[1], fv = p[2];
    const double x = X[0], y = X[1], z = X[2];
    double xx = x * x;
    double yy = y * y;
    double zz = z * z;
    double rho = sqrt(xx + yy + zz);
    double deninv = 1. / (xi * rho + z);
    double xn = x * deninv;
    double yn = y * deninv;
    du[0] = -fu * xn * deninv * rho;
    du[1] = xn;
    du[2] = 0;
    du[3] = 1;
    du[4] = 0;
    dv[0] = -fv * yn * deninv * rho;
    dv[1] = 0;
    dv[2] = yn;
    dv[3] = 0;
    dv[4] = 1;
    return 1;
}

/* MeiProjector  include/projection/mei.h:29-68 (never reports failure) */
static int mei_project(const double *p, const double X[3], double uv[2])
{
    const double xi = p[0], k1 = p[1], k2 = p[2], k3 = p[3], k4 = p[4], k5 = p[5];
    const double fu = p[6], fv = p[7], u0 = p[8], v0 = p[9];
    const double x = X[0], y = X[1], z = X[2];
    double rho = sqrt(z * z + x * x + y * y);
    double denominv = 1. / (z + xi * rho);
    double xn = x * denominv;
    double yn = y * denominv;
    double xx = xn * xn, xy = xn * yn, yy = yn * yn;
    double r2 = xx + yy;
    double D = 1. + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
    double deltax = 2. * k4 * xy + k5 * (r2 + 2. * xx);
    double deltay = 2. * k5 * xy + k4 * (r2 + 2. * yy);
    uv[0] = fu * (xn * D + deltax) + u0;
    uv[1] = fv * (yn * D + deltay) + v0;
    return 1;
}

/* distortion Jacobian rows shared by mei.h:158-184 and mei.h:236-247 ; a = fu*dudmn, b = fv*dvdmn */
static void mei_distortion_rows(const double *p, double xn, double yn, double a[2], double b[2],
                                double *r2_, double *D_)
{
    const double k1 = p[1], k2 = p[2], k3 = p[3], k4 = p[4], k5 = p[5], fu = p[6], fv = p[7];
    double xxn = xn * xn;
    double yyn = yn * yn;
    double xyn = xn * yn;
    double r2 = yn * yn + xn * xn;
    double D = 1. + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
    double dDdr2 = k1 + 2 * k2 * r2 + 3 * k3 * r2 * r2;
    a[0] = D + 2 * xxn * dDdr2 + 2 * k4 * yn + 6 * k5 * xn;
    a[1] = 2 * xyn * dDdr2 + 2 * k4 * xn + 2 * k5 * yn;
    b[0] = 2 * xyn * dDdr2 + 2 * k5 * yn + 2 * k4 * xn;
    b[1] = D + 2 * yyn * dDdr2 + 2 * k5 * xn + 6 * k4 * yn;
    a[0] *= fu; a[1] *= fu;
    b[0] *= fv; b[1] *= fv;
    *r2_ = r2;
    *D_ = D;
}

/* MeiCamera::projectionJacobian  include/projection/mei.h:121-191 */
static int mei_projection_jacobian(const double *p, const double X[3], double dudx[3], double dvdx[3])
{
    double dm[6], xn, yn, rho, deninv, a[2], b[2], r2, D;
    unified_dm(p[0], X, dm, &xn, &yn, &rho, &deninv);
    mei_distortion_rows(p, xn, yn, a, b, &r2, &D);
    for (int j = 0; j < 3; j++) {
        dudx[j] = a[0] * dm[j] + a[1] * dm[3 + j]; /* dudmn * jac_m  (1x2 . 2x3) */
        dvdx[j] = b[0] * dm[j] + b[1] * dm[3 + j];
    }
    return 1;
}

/* MeiCamera::intrinsicJacobian  include/projection/mei.h:193-285 */
static int mei_intrinsic_jacobian(const double *p, const double X[3], double *du, double *dv)
{
    const double xi = p[0], k4 = p[4], k5 = p[5], fu = p[6], fv = p[7];
    const double x = X[0], y = X[1], z = X[2];
    double xx = x * x;
    double yy = y * y;
    double zz = z * z;
    double rho = sqrt(xx + yy + zz);
    double deninv = 1. / (xi * rho + z);
    double xn = x * deninv;
    double yn = y * deninv;
    double xxn = xn * xn;
    double yyn = yn * yn;
    double xyn = xn * yn;
    double a[2], b[2], r2, D;
    mei_distortion_rows(p, xn, yn, a, b, &r2, &D);
    double deltax = 2. * k4 * xyn + k5 * (r2 + 2. * xxn);
    double deltay = 2. * k5 * xyn + k4 * (r2 + 2. * yyn);
    double xd = xn * D + deltax;
    double yd = yn * D + deltay;
    double dxndxi = -xn * deninv * rho;
    double dyndxi = -yn * deninv * rho;

    du[0] = a[0] * dxndxi + a[1] * dyndxi;
    du[1] = fu * xn * r2;
    du[2] = fu * xn * r2 * r2;
    du[3] = fu * xn * r2 * r2 * r2;
    du[4] = 2. * fu * xyn;
    du[5] = fu * (r2 + 2. * xxn);
    du[6] = xd;
    du[7] = 0;
    du[8] = 1;
    du[9] = 0;

    dv[0] = b[0] * dxndxi + b[1] * dyndxi;
    dv[1] = fv * yn * r2;
    dv[2] = fv * yn * r2 * r2;
    dv[3] = fv * yn * r2 * r2 * r2;
    dv[4] = fv * (r2 + 2. * yyn);
    dv[5] = 2. * fv * xyn;
    dv[6] = 0;
    dv[7] = yd;
    dv[8] = 0;
    dv[9] = 1;
    return 1;
}

/* ICamera virtual dispatch  include/projection/generic_camera.h:39-51 */
int vgo_project_point(int model, const double *intr, const double X[3], double uv[2])
{
    switch (model) {
    case VGO_MODEL_EUCM: return eucm_project(intr, X, uv);
    case VGO_MODEL_UCM: return ucm_project(intr, X, uv);
    case VGO_MODEL_MEI: return mei_project(intr, X, uv);
    default: return 0;
    }
}

int vgo_projection_jacobian(int model, const double *intr, const double X[3], double dudx[3], double dvdx[3])
{
    switch (model) {
    case VGO_MODEL_EUCM: return eucm_projection_jacobian(intr, X, dudx, dvdx);
    case VGO_MODEL_UCM: return ucm_projection_jacobian(intr, X, dudx, dvdx);
    case VGO_MODEL_MEI: return mei_projection_jacobian(intr, X, dudx, dvdx);
    default: return 0;
    }
}

int vgo_intrinsic_jacobian(int model, const double *intr, const double X[3], double *du, double *dv)
{
    switch (model) {
    case VGO_MODEL_EUCM: return eucm_intrinsic_jacobian(intr, X, du, dv);
    case VGO_MODEL_UCM: return ucm_intrinsic_jacobian(intr, X, du, dv);
    case VGO_MODEL_MEI: return mei_intrinsic_jacobian(intr, X, du, dv);
    default: return 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * InterJacobian  include/projection/jacobian.h:136-171
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double R12[9], M12[9], t13[3];
} inter_jacobian;

/* ctor  jacobian.h:139-152 */
static void inter_jacobian_init(inter_jacobian *ij, const double xi13[6], const double xi23[6], int inverted)
{
    double Ra[9], Rb[9], negr[3] = {-xi23[3], -xi23[4], -xi23[5]}, M[9];
    vgo_rotation_matrix(xi13 + 3, Ra); /* xi13.rotMat()    transformation.h:131 */
    vgo_rotation_matrix(negr, Rb);     /* xi23.rotMatInv() transformation.h:132 */
    mat3_mul(Ra, Rb, ij->R12);
    ij->t13[0] = xi13[0]; ij->t13[1] = xi13[1]; ij->t13[2] = xi13[2];
    vgo_inter_omega_rot(xi23 + 3, M);
    mat3_mul(ij->R12, M, ij->M12);
    if (inverted) {
        for (int i = 0; i < 9; i++) { ij->R12[i] *= -1; ij->M12[i] *= -1; }
    }
}

/* dpdxi  jacobian.h:155-171 ; the camera's return value is ignored (jacobian.h:158) */
static void inter_jacobian_dpdxi(const inter_jacobian *ij, int model, const double *intr,
                                 const double X1[3], double *dudxi, double *dvdxi)
{
    double P[6];
    vgo_projection_jacobian(model, intr, X1, P, P + 3);
    double t3X[3] = {X1[0] - ij->t13[0], X1[1] - ij->t13[1], X1[2] - ij->t13[2]};
    double H[9];
    hat_(t3X, H);
    for (int row = 0; row < 2; row++) {
        const double *p = P + 3 * row;
        double *out = row == 0 ? dudxi : dvdxi;
        /* dudtr = projJac.row * R12 */
        for (int j = 0; j < 3; j++)
            out[j] = p[0] * ij->R12[0 + j] + p[1] * ij->R12[3 + j] + p[2] * ij->R12[6 + j];
        /* dudrot = ((-projJac.row) * hat(t3X)) * M12 */
        double n[3] = {-p[0], -p[1], -p[2]}, tmp[3];
        for (int j = 0; j < 3; j++) tmp[j] = n[0] * H[0 + j] + n[1] * H[3 + j] + n[2] * H[6 + j];
        for (int j = 0; j < 3; j++)
            out[3 + j] = tmp[0] * ij->M12[0 + j] + tmp[1] * ij->M12[3 + j] + tmp[2] * ij->M12[6 + j];
    }
}

/* ------------------------------------------------------------------------------------------
 * GenericProjectionJac::Evaluate  src/calibration/calib_cost_functions.cpp:28-117
 * ---------------------------------------------------------------------------------------- */
#define VGO_STACK_POINTS 256

int vgo_eval_block(int model, int L, const int *status, int N, const double *grid, const double *obs,
                   const double *const *params, double *residual, double **jac)
{
    const int K = vgo_num_intrinsics(model);
    if (K < 0 || L < 0 || L > VGO_MAX_CHAIN || N < 0 || !params || !residual) return 0;

    /* :32-46  xiAcc = Id; compose / composeInverse each chain member */
    double xiAcc[6] = {0, 0, 0, 0, 0, 0}; /* Transformation() = zeros  transformation.h:36 */
    for (int l = 0; l < L; l++) {
        double nxt[6];
        if (status[l] == VGO_TRANSFORM_DIRECT) vgo_compose(xiAcc, params[1 + l], nxt);
        else if (status[l] == VGO_TRANSFORM_INVERSE) vgo_compose_inverse(xiAcc, params[1 + l], nxt);
        else return 0;
        memcpy(xiAcc, nxt, sizeof nxt);
    }

    /* :49-50  pointCam = R(xiAcc.rot) * grid + xiAcc.trans   transformation.h:147-155,180-188 */
    double R[9];
    vgo_rotation_matrix(xiAcc + 3, R);

    const double *intr = params[0]; /* :53-54 setParameters */

    /* the reference keeps pointCamVec; recomputing X per use gives bit-identical values */
#define POINT_CAM(i, X)                                                                            \
    do {                                                                                           \
        const double *g_ = grid + 3 * (i);                                                         \
        (X)[0] = (R[0] * g_[0] + R[1] * g_[1] + R[2] * g_[2]) + xiAcc[0];                          \
        (X)[1] = (R[3] * g_[0] + R[4] * g_[1] + R[5] * g_[2]) + xiAcc[1];                          \
        (X)[2] = (R[6] * g_[0] + R[7] * g_[1] + R[8] * g_[2]) + xiAcc[2];                          \
    } while (0)

    /* :57-71 residuals */
    for (int i = 0; i < N; i++) {
        double X[3], uv[2];
        POINT_CAM(i, X);
        if (vgo_project_point(model, intr, X, uv)) {
            residual[2 * i] = uv[0] - obs[2 * i];
            residual[2 * i + 1] = uv[1] - obs[2 * i + 1];
        } else {
            residual[2 * i] = VGO_DOUBLE_BIG;
            residual[2 * i + 1] = VGO_DOUBLE_BIG;
        }
    }

    if (jac != NULL) {
        /* :76-103 second chain walk */
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (int l = 0; l < L; l++) {
            const double *xi23 = params[1 + l];
            double xi13[6], nxt[6];
            int inverted = status[l] == VGO_TRANSFORM_INVERSE;
            if (!inverted) {
                vgo_compose(acc, xi23, nxt);
                memcpy(acc, nxt, sizeof nxt);
                memcpy(xi13, acc, sizeof acc);
            } else {
                memcpy(xi13, acc, sizeof acc);
                vgo_compose_inverse(acc, xi23, nxt);
                memcpy(acc, nxt, sizeof nxt);
            }
            if (jac[1 + l] != NULL) {
                inter_jacobian ij;
                inter_jacobian_init(&ij, xi13, xi23, inverted);
                for (int i = 0; i < N; i++) {
                    double X[3];
                    POINT_CAM(i, X);
                    inter_jacobian_dpdxi(&ij, model, intr, X, jac[1 + l] + i * 12, jac[1 + l] + i * 12 + 6);
                }
            }
        }
        /* :105-114 intrinsic Jacobian */
        if (jac[0] != NULL) {
            for (int i = 0; i < N; i++) {
                double X[3];
                POINT_CAM(i, X);
                vgo_intrinsic_jacobian(model, intr, X, jac[0] + (size_t)i * 2 * K, jac[0] + ((size_t)i * 2 + 1) * K);
            }
        }
    }
#undef POINT_CAM
    return 1; /* :116 */
}

long vgo_eval_dataset(int model, int L, const int *status, int N, const double *grid, long n_blocks,
                      const double *obs, const double *param_vec, long intr_offset,
                      const long *member_base, const long *member_stride, const long *seq_index,
                      double *residuals, double *jac_intr, double *const *jac_member, int threads)
{
    const int K = vgo_num_intrinsics(model);
    if (K < 0 || L < 0 || L > VGO_MAX_CHAIN) return 0;
    long done = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : done) num_threads(threads > 1 ? threads : 1)
#endif
    for (long b = 0; b < n_blocks; b++) {
        const double *params[1 + VGO_MAX_CHAIN];
        double *jac[1 + VGO_MAX_CHAIN];
        params[0] = param_vec + intr_offset;
        jac[0] = jac_intr ? jac_intr + (size_t)b * 2 * N * K : NULL;
        for (int l = 0; l < L; l++) {
            params[1 + l] = param_vec + member_base[l] + member_stride[l] * seq_index[b];
            jac[1 + l] = (jac_member && jac_member[l]) ? jac_member[l] + (size_t)b * 2 * N * 6 : NULL;
        }
        int want_jac = jac_intr != NULL || jac_member != NULL;
        done += vgo_eval_block(model, L, status, N, grid, obs + (size_t)b * 2 * N, params,
                               residuals + (size_t)b * 2 * N, want_jac ? jac : NULL);
    }
    return done;
}

/* column c of the stacked row block [J_0 | J_1 .. J_L | r], row `row` */
static double stacked_entry(int K, int L, int row, int c, const double *residual,
                            const double *jac_intr, const double *const *jac_member)
{
    if (c < K) return jac_intr[(size_t)row * K + c];
    c -= K;
    if (c < 6 * L) return jac_member[c / 6][(size_t)row * 6 + c % 6];
    return residual[row];
}

void vgo_block_gram(int K, int L, int N, const double *residual, const double *jac_intr,
                    const double *const *jac_member, double *gram)
{
    const int W = K + 6 * L + 1;
    for (int a = 0; a < W; a++)
        for (int b = a; b < W; b++) {
            long double s = 0.0L;
            for (int row = 0; row < 2 * N; row++)
                s += (long double)stacked_entry(K, L, row, a, residual, jac_intr, jac_member) *
                     (long double)stacked_entry(K, L, row, b, residual, jac_intr, jac_member);
            gram[a * W + b] = (double)s;
            gram[b * W + a] = (double)s;
        }
}

void vgo_block_gram_fast(int K, int L, int N, const double *residual, const double *jac_intr,
                         const double *const *jac_member, double *gram)
{
    const int W = K + 6 * L + 1;
    double row_buf[10 + 6 * VGO_MAX_CHAIN + 1];
    for (int i = 0; i < W * W; i++) gram[i] = 0.;
    for (int row = 0; row < 2 * N; row++) {
        for (int c = 0; c < W; c++) row_buf[c] = stacked_entry(K, L, row, c, residual, jac_intr, jac_member);
        for (int a = 0; a < W; a++) {
            const double va = row_buf[a];
            for (int b = a; b < W; b++) gram[a * W + b] += va * row_buf[b];
        }
    }
    for (int a = 0; a < W; a++)
        for (int b = a + 1; b < W; b++) gram[b * W + a] = gram[a * W + b];
}

long vgo_dataset_gram(int K, int L, int N, long n_blocks, const double *residuals, const double *jac_intr,
                      const double *const *jac_member, double *grams, double *sum, int threads)
{
    const int W = K + 6 * L + 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 1 ? threads : 1)
#endif
    for (long b = 0; b < n_blocks; b++) {
        const double *jm[VGO_MAX_CHAIN];
        for (int l = 0; l < L; l++) jm[l] = jac_member[l] + (size_t)b * 2 * N * 6;
        vgo_block_gram_fast(K, L, N, residuals + (size_t)b * 2 * N, jac_intr + (size_t)b * 2 * N * K, jm,
                            grams + (size_t)b * W * W);
    }
    if (sum) {
        for (int e = 0; e < W * W; e++) sum[e] = 0.;
        for (long b = 0; b < n_blocks; b++)
            for (int e = 0; e < W * W; e++) sum[e] += grams[(size_t)b * W * W + e];
    }
    return n_blocks;
}

/* TransformationPrior: ctor include/calibration/calib_cost_functions.h:79-103, Evaluate src/.../calib_cost_functions.cpp:214-228 */
static void inverse_compose(const double a[6], const double b[6], double out[6]) /* transformation.h:90-99 */
{
    double q1[4], q2[4], q1inv[4], qres[4];
    vgo_quat_from_rotvec(a + 3, q1);
    vgo_quat_from_rotvec(b + 3, q2);
    q1inv[0] = -q1[0]; q1inv[1] = -q1[1]; q1inv[2] = -q1[2]; q1inv[3] = q1[3];
    double d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    quat_rotate(q1inv, d, out);
    quat_mul(q1inv, q2, qres);
    vgo_quat_to_rotvec(qres, out + 3);
}

void vgo_transformation_prior(const double stiffness[6], const double xi_prior[6], const double xi[6],
                              double residual[6], double jac[36])
{
    double A[36], R[9], M[9];
    for (int i = 0; i < 36; i++) A[i] = 0.;
    for (int i = 0; i < 6; i++) A[6 * i + i] = stiffness[i];
    vgo_rotation_matrix(xi_prior + 3, R);
    vgo_inter_omega_rot(xi_prior + 3, M);
    /* _A.bottomRightCorner<3,3>() = _A.bottomRightCorner<3,3>() * M */
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0.;
            for (int k = 0; k < 3; k++) s += (r == k ? stiffness[3 + r] : 0.) * M[3 * k + c];
            A[6 * (3 + r) + 3 + c] = s;
        }
    double err[6], e2[6];
    inverse_compose(xi_prior, xi, err);
    for (int i = 0; i < 3; i++) {
        e2[i] = R[3 * i] * err[0] + R[3 * i + 1] * err[1] + R[3 * i + 2] * err[2];
        e2[3 + i] = R[3 * i] * err[3] + R[3 * i + 1] * err[4] + R[3 * i + 2] * err[5];
    }
    for (int i = 0; i < 6; i++) {
        double s = 0.;
        for (int k = 0; k < 6; k++) s += A[6 * i + k] * e2[k];
        residual[i] = s;
    }
    if (jac) memcpy(jac, A, sizeof A);
}

/* ------------------------------------------------------------------------------------------
 * OdometryPrior (calibration version): ctor src/calibration/calib_cost_functions.cpp:119-167,
 * Evaluate :171-212, declaration include/calibration/calib_cost_functions.h:64-77
 * ---------------------------------------------------------------------------------------- */
static void mat3_inverse(const double M[9], double I[9]) /* Eigen's 3x3 inverse: adjugate / determinant */
{
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
    const double id = 1. / det;
    I[0] = c00 * id; I[1] = (M[2] * M[7] - M[1] * M[8]) * id; I[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    I[3] = c01 * id; I[4] = (M[0] * M[8] - M[2] * M[6]) * id; I[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    I[6] = c02 * id; I[7] = (M[1] * M[6] - M[0] * M[7]) * id; I[8] = (M[0] * M[4] - M[1] * M[3]) * id;
}

void vgo_odometry_prior_init(double errV, double errW, double lambda, const double xi1[6], const double xi2[6],
                             double zetaPrior[6], double A[36])
{
    inverse_compose(xi1, xi2, zetaPrior); /* _zetaPrior(xi1.inverseCompose(xi2)) */
    const double MIN_SIGMA_V = 0.01, MIN_SIGMA_W = 0.01, MIN_DELTA = 0.01, MIN_L = 0.01;
    const double nr = norm3(zetaPrior + 3), nt = norm3(zetaPrior);
    const double delta = nr > MIN_DELTA ? nr : MIN_DELTA;
    const double l = nt > MIN_L ? nt : MIN_L;
    const double delta2 = delta / 2., l2 = l / 2.;
    const double s = sin(delta2), c = cos(delta2);
    const double dfdu[6] = {c, l2 * s, -s, l2 * c, 0, 1}; /* 3x2 row-major */
    double Cu0 = errV * errV * l * l, Cu1 = errW * errW * delta * delta;
    if (Cu0 < MIN_SIGMA_V * MIN_SIGMA_V) Cu0 = MIN_SIGMA_V * MIN_SIGMA_V;
    if (Cu1 < MIN_SIGMA_W * MIN_SIGMA_W) Cu1 = MIN_SIGMA_W * MIN_SIGMA_W;
    double Cx[9], CxInv[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            Cx[3 * i + j] = dfdu[2 * i] * Cu0 * dfdu[2 * j] + dfdu[2 * i + 1] * Cu1 * dfdu[2 * j + 1] + (i == j ? lambda * lambda : 0.);
    mat3_inverse(Cx, CxInv);
    /* Eigen::LLT<Matrix3d>(CxInv).matrixU(): CxInv = L L^T, U = L^T */
    double L[9] = {0};
    for (int r = 0; r < 3; r++)
        for (int cc = 0; cc <= r; cc++) {
            double v = CxInv[3 * r + cc];
            for (int k = 0; k < cc; k++) v -= L[3 * r + k] * L[3 * cc + k];
            L[3 * r + cc] = (r == cc) ? sqrt(v) : v / L[3 * cc + cc];
        }
    double U[9];
    for (int r = 0; r < 3; r++)
        for (int cc = 0; cc < 3; cc++) U[3 * r + cc] = L[3 * cc + r];
    for (int i = 0; i < 36; i++) A[i] = 0.;
    A[0] = U[0]; A[1] = U[1]; A[6] = U[3]; A[7] = U[4]; /* _A.topLeftCorner<2,2>() = U.topLeftCorner<2,2>() */
    A[5] = U[2]; A[11] = U[5];                          /* _A.topRightCorner<2,1>() = U.topRightCorner<2,1>(): column 5 of the 6x6 */
    A[14] = 1. / lambda;                                /* _A(2,2) */
    A[21] = 1. / lambda; A[28] = 1. / lambda; A[35] = U[8]; /* bottomRightCorner<3,3>() = diag(1/lambda, 1/lambda, U(2,2)) */
}

/* 6x6 row-major blocks: out = [[Ra, 0],[0, Rb]] */
static void blockdiag6(const double Ra[9], const double Rb[9], double out[36])
{
    for (int i = 0; i < 36; i++) out[i] = 0.;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            out[6 * i + j] = Ra[3 * i + j];
            out[6 * (3 + i) + 3 + j] = Rb[3 * i + j];
        }
}

static void mat6_mul(const double A[36], const double B[36], double C[36])
{
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double s = 0.;
            for (int k = 0; k < 6; k++) s += A[6 * i + k] * B[6 * k + j];
            C[6 * i + j] = s;
        }
}

void vgo_odometry_prior_eval(const double zetaPrior[6], const double A[36], const double xi1[6], const double xi2[6],
                             double residual[6], double J1[36], double J2[36])
{
    double zeta[6], err[6];
    inverse_compose(xi1, xi2, zeta);
    inverse_compose(zetaPrior, zeta, err);
    for (int i = 0; i < 6; i++) {
        double s = 0.;
        for (int k = 0; k < 6; k++) s += A[6 * i + k] * err[k];
        residual[i] = s;
    }
    if (J1) {
        double n1[3] = {-xi1[3], -xi1[4], -xi1[5]}, R10[9], M[9], RM[9], J1m[36];
        vgo_rotation_matrix(n1, R10);      /* xi1.rotMatInv() */
        vgo_inter_omega_rot(xi1 + 3, M);
        mat3_mul(R10, M, RM);              /* R10 * interOmegaRot(xi1.rot()) */
        blockdiag6(R10, RM, J1m);
        /* TT = zeta.screwTransfInv(): [[R, -R hat(t)],[0, R]], R = zeta.rotMatInv()   transformation.h:234-243 */
        double nz[3] = {-zeta[3], -zeta[4], -zeta[5]}, R[9], H[9], RH[9], TT[36];
        vgo_rotation_matrix(nz, R);
        hat_(zeta, H);
        mat3_mul(R, H, RH);
        for (int i = 0; i < 36; i++) TT[i] = 0.;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                TT[6 * i + j] = R[3 * i + j];
                TT[6 * i + 3 + j] = -RH[3 * i + j];
                TT[6 * (3 + i) + 3 + j] = R[3 * i + j];
            }
        double T1[36], T2[36];
        mat6_mul(A, TT, T1);
        mat6_mul(T1, J1m, T2);
        for (int i = 0; i < 36; i++) J1[i] = -T2[i]; /* jac = -_A * TT * J1 */
    }
    if (J2) {
        double n2[3] = {-xi2[3], -xi2[4], -xi2[5]}, R20[9], M[9], RM[9], J2m[36];
        vgo_rotation_matrix(n2, R20);
        vgo_inter_omega_rot(xi2 + 3, M);
        mat3_mul(R20, M, RM);
        blockdiag6(R20, RM, J2m);
        mat6_mul(A, J2m, J2); /* jac = _A * J2 */
    }
}

/* ------------------------------------------------------------------------------------------
 * OdometryCost: differential-drive odometry with wheel radii / track gauge as a third parameter block.
 * src/calibration/odometry_cost_function.cpp: odom_zeta_i :10-36, zeta_i_jacobian :39-69, tf0n_jac_calc :72-94,
 * calc_acc :96-144, ctor :147-197, Evaluate :202-266.  Parameter blocks (6, 6, 3) as it is ADDED to the problem
 * (src/calibration/unified_calibration.cpp:732-735); the class declares one block only.
 * ---------------------------------------------------------------------------------------- */
static void compose_(const double a[6], const double b[6], double out[6]) /* transformation.h:80-88 */
{
    double q1[4], q2[4], qres[4], rt[3];
    vgo_quat_from_rotvec(a + 3, q1);
    vgo_quat_from_rotvec(b + 3, q2);
    quat_rotate(q1, b, rt);
    out[0] = rt[0] + a[0]; out[1] = rt[1] + a[1]; out[2] = rt[2] + a[2];
    quat_mul(q1, q2, qres);
    vgo_quat_to_rotvec(qres, out + 3);
}

static void inverse_(const double a[6], double out[6]) /* transformation.h:112-119: t = -(R(-r) t), r = -r */
{
    double n[3] = {-a[3], -a[4], -a[5]}, R[9];
    vgo_rotation_matrix(n, R);
    for (int i = 0; i < 3; i++) out[i] = (-R[3 * i]) * a[0] + (-R[3 * i + 1]) * a[1] + (-R[3 * i + 2]) * a[2];
    out[3] = n[0]; out[4] = n[1]; out[5] = n[2];
}

/* tf0n_jac_calc: the chain 0T1 ... 0Tn of the wheel increments and d(zeta_i)/d(intrinsics) of every step */
static void odo_chain(int n, const double *deltaQ, const double intr[3], double *tf0 /*[n][6]*/, double *jz /*[n][9]*/)
{
    const double r1 = intr[0], r2 = intr[1], g = intr[2];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        const double dl = deltaQ[2 * i], dr = deltaQ[2 * i + 1];
        /* odom_zeta_i: v_w = [[r1/2, r2/2], [-(r1/g), r2/g]] * delta; zeta_i = (v, 0, w) */
        const double v = (r1 / 2) * dl + (r2 / 2) * dr;
        const double w = (-(r1 / g)) * dl + (r2 / g) * dr;
        const double step[6] = {v, 0., 0., 0., 0., w};
        double nxt[6];
        compose_(acc, step, nxt);
        memcpy(acc, nxt, sizeof acc);
        memcpy(tf0 + 6 * i, acc, sizeof acc);
        double *J = jz + 9 * i; /* zeta_i_jacobian */
        J[0] = dl / 2; J[1] = dr / 2; J[2] = 0;
        J[3] = 0; J[4] = 0; J[5] = 0;
        J[6] = -dl / g; J[7] = dr / g; J[8] = (r1 * dl - r2 * dr) / (g * g);
    }
}

void vgo_odometry_cost_init(double errV, double errW, double lambda, int n, const double *deltaQ, const double intr_prior[3],
                            double zetaPrior[6], double A[36])
{
    double *tf0 = (double *)malloc(sizeof(double) * 6 * (size_t)n), *jz = (double *)malloc(sizeof(double) * 9 * (size_t)n);
    odo_chain(n, deltaQ, intr_prior, tf0, jz);
    memcpy(zetaPrior, tf0 + 6 * (n - 1), 6 * sizeof(double));
    free(tf0);
    free(jz);
    /* the rest of the constructor (:160-194) repeats OdometryPrior's (calib_cost_functions.cpp:127-167) on this zetaPrior */
    {
        const double MIN_SIGMA_V = 0.01, MIN_SIGMA_W = 0.01, MIN_DELTA = 0.01, MIN_L = 0.01;
        const double nr = norm3(zetaPrior + 3), nt = norm3(zetaPrior);
        const double delta = nr > MIN_DELTA ? nr : MIN_DELTA, l = nt > MIN_L ? nt : MIN_L;
        const double s = sin(delta / 2.), c = cos(delta / 2.), l2 = l / 2.;
        const double dfdu[6] = {c, l2 * s, -s, l2 * c, 0, 1};
        double Cu0 = errV * errV * l * l, Cu1 = errW * errW * delta * delta;
        if (Cu0 < MIN_SIGMA_V * MIN_SIGMA_V) Cu0 = MIN_SIGMA_V * MIN_SIGMA_V;
        if (Cu1 < MIN_SIGMA_W * MIN_SIGMA_W) Cu1 = MIN_SIGMA_W * MIN_SIGMA_W;
        double Cx[9], CxInv[9], L[9] = {0}, U[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                Cx[3 * i + j] = dfdu[2 * i] * Cu0 * dfdu[2 * j] + dfdu[2 * i + 1] * Cu1 * dfdu[2 * j + 1] + (i == j ? lambda * lambda : 0.);
        mat3_inverse(Cx, CxInv);
        for (int r = 0; r < 3; r++)
            for (int cc = 0; cc <= r; cc++) {
                double v = CxInv[3 * r + cc];
                for (int k = 0; k < cc; k++) v -= L[3 * r + k] * L[3 * cc + k];
                L[3 * r + cc] = (r == cc) ? sqrt(v) : v / L[3 * cc + cc];
            }
        for (int r = 0; r < 3; r++)
            for (int cc = 0; cc < 3; cc++) U[3 * r + cc] = L[3 * cc + r];
        for (int i = 0; i < 36; i++) A[i] = 0.;
        A[0] = U[0]; A[1] = U[1]; A[6] = U[3]; A[7] = U[4];
        A[5] = U[2]; A[11] = U[5];
        A[14] = 1. / lambda;
        A[21] = 1. / lambda; A[28] = 1. / lambda; A[35] = U[8];
    }
}

void vgo_odometry_cost_eval(const double A[36], int n, const double *deltaQ, const double xi1[6], const double xi2[6],
                            const double intr[3], double residual[6], double J1[36], double J2[36], double J3[18])
{
    double zeta[6], err[6];
    inverse_compose(xi1, xi2, zeta);
    double *tf0 = (double *)malloc(sizeof(double) * 6 * (size_t)n), *jz = (double *)malloc(sizeof(double) * 9 * (size_t)n);
    odo_chain(n, deltaQ, intr, tf0, jz);
    const double *zeta_odo = tf0 + 6 * (n - 1);
    double delta[6];
    inverse_compose(zeta_odo, zeta, delta); /* zeta_odo.inverseCompose(zeta) */
    memcpy(err, delta, sizeof err);
    for (int i = 0; i < 6; i++) {
        double s = 0.;
        for (int k = 0; k < 6; k++) s += A[6 * i + k] * err[k];
        residual[i] = s;
    }
    if (J1 || J2) { /* identical to OdometryPrior's two pose blocks (:231-252 == calib_cost_functions.cpp:187-209) */
        double r_[6];
        const double zero[6] = {0, 0, 0, 0, 0, 0};
        vgo_odometry_prior_eval(zero, A, xi1, xi2, r_, J1, J2);
    }
    if (J3) {
        /* calc_acc: ACC = sum_i R(0T(i-1)) * [[1,0,-t_y],[0,1,t_x],[0,0,1]] * jac_zeta_i, t = trans(iTn) */
        double ACC[9] = {0};
        for (int i = 0; i < n; i++) {
            const double zero[6] = {0, 0, 0, 0, 0, 0};
            const double *tf0j = i > 0 ? tf0 + 6 * (i - 1) : zero;
            double R0j[9], ti0[6], tin[6];
            vgo_rotation_matrix(tf0j + 3, R0j);
            inverse_(tf0 + 6 * i, ti0);
            compose_(ti0, zeta_odo, tin);
            const double Jm[9] = {1, 0, -tin[1], 0, 1, tin[0], 0, 0, 1};
            double T[9], T2[9];
            mat3_mul(R0j, Jm, T);
            mat3_mul(T, jz + 9 * i, T2);
            for (int k = 0; k < 9; k++) ACC[k] = ACC[k] + T2[k];
        }
        double acc63[18] = {ACC[0], ACC[1], ACC[2], ACC[3], ACC[4], ACC[5], 0, 0, 0, 0, 0, 0, 0, 0, 0, ACC[6], ACC[7], ACC[8]};
        /* jac = -_A * delta.screwTransfInv() * [[R31, 0], [0, R31 M(zeta_odo.rot)]] * jac_intrinsic */
        double n3[3] = {-zeta_odo[3], -zeta_odo[4], -zeta_odo[5]}, R31[9], M[9], RM[9], J3m[36];
        vgo_rotation_matrix(n3, R31);
        vgo_inter_omega_rot(zeta_odo + 3, M);
        mat3_mul(R31, M, RM);
        blockdiag6(R31, RM, J3m);
        double nd[3] = {-delta[3], -delta[4], -delta[5]}, R[9], H[9], RH[9], TT[36];
        vgo_rotation_matrix(nd, R);
        hat_(delta, H);
        mat3_mul(R, H, RH);
        for (int i = 0; i < 36; i++) TT[i] = 0.;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                TT[6 * i + j] = R[3 * i + j];
                TT[6 * i + 3 + j] = -RH[3 * i + j];
                TT[6 * (3 + i) + 3 + j] = R[3 * i + j];
            }
        double T1[36], T2[36];
        mat6_mul(A, TT, T1);
        mat6_mul(T1, J3m, T2);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0.;
                for (int k = 0; k < 6; k++) s += T2[6 * i + k] * acc63[3 * k + j];
                J3[3 * i + j] = -s;
            }
    }
    free(tf0);
    free(jz);
}

/* ------------------------------------------------------------------------------------------
 * Localization costs on the same camera models:
 *   CameraJacobian            include/projection/jacobian.h:51-119
 *   Triangulator              include/reconstruction/triangulator.h:31-38, src/reconstruction/triangulator.cpp:114-259
 *   MonoReprojectCost         include/localization/local_cost_functions.h:159-180, src/localization/local_cost_functions.cpp:216-278
 *   SparseReprojectCost       .h:183-208, .cpp:281-391
 * ---------------------------------------------------------------------------------------- */

/* CameraJacobian ctors, jacobian.h:54-71.  T23 == NULL: the one-transform ctor. */
void vgo_camera_jacobian_init(const double T12[6], const double *T23, double L11[9], double L12[9], double L22[9])
{
    double n12[3] = {-T12[3], -T12[4], -T12[5]}, M[9];
    vgo_inter_omega_rot(T12 + 3, M);
    if (T23) {
        double R21[9], R32[9], n23[3] = {-T23[3], -T23[4], -T23[5]};
        vgo_rotation_matrix(n12, R21); /* T12.rotMatInv() */
        vgo_rotation_matrix(n23, R32); /* T23.rotMatInv() */
        mat3_mul(R32, R21, L11);       /* L11 = R32 * R21 */
        mat3_mul(L11, M, L22);         /* L22 = L11 * M   */
        /* L12 = -R32 * hat(T23.trans()) * R21 * M : left to right */
        double nR32[9], H[9], a[9], b[9];
        for (int i = 0; i < 9; i++) nR32[i] = -R32[i];
        hat_(T23, H);
        mat3_mul(nR32, H, a);
        mat3_mul(a, R21, b);
        mat3_mul(b, M, L12);
    } else {
        vgo_rotation_matrix(n12, L11); /* L11( T12.rotMatInv() ) */
        for (int i = 0; i < 9; i++) L12[i] = 0.; /* L12( Matrix3d::Zero() ) */
        mat3_mul(L11, M, L22);         /* L22( L11 * interOmegaRot(T12.rot()) ) */
    }
}

/* CameraJacobian::dpdxi jacobian.h:75-96 and ::dfdxi :99-113 (grad != NULL).  two = twoTransforms. */
void vgo_camera_jacobian_eval(int model, const double *intr, int two, const double L11[9], const double L12[9],
                              const double L22[9], const double X2[3], const double *grad /* [2] or NULL */,
                              double *dudxi, double *dvdxi, double *dfdxi)
{
    double P[6];
    const int ok = vgo_projection_jacobian(model, intr, X2, P, P + 3);
    if (!ok) {
        if (dudxi) for (int i = 0; i < 6; i++) dudxi[i] = 0.;
        if (dvdxi) for (int i = 0; i < 6; i++) dvdxi[i] = 0.;
        if (dfdxi) for (int i = 0; i < 6; i++) dfdxi[i] = 0.;
        return;
    }
    double H[9], B[9];
    hat_(X2, H);
    mat3_mul(H, L22, B); /* hat(X2) * L22 */
    if (two) for (int i = 0; i < 9; i++) B[i] = B[i] - L12[i];
    for (int row = 0; row < 2; row++) {
        double *out = row == 0 ? dudxi : dvdxi;
        if (!out) continue;
        const double *p = P + 3 * row;
        const double n[3] = {-p[0], -p[1], -p[2]};
        for (int j = 0; j < 3; j++) out[j] = n[0] * L11[0 + j] + n[1] * L11[3 + j] + n[2] * L11[6 + j]; /* -row * L11 */
        for (int j = 0; j < 3; j++) out[3 + j] = p[0] * B[0 + j] + p[1] * B[3 + j] + p[2] * B[6 + j];   /* row * B */
    }
    if (dfdxi && grad) {
        double d[3], n[3];
        for (int j = 0; j < 3; j++) d[j] = grad[0] * P[j] + grad[1] * P[3 + j]; /* dfdX = grad * projJac */
        for (int j = 0; j < 3; j++) n[j] = -d[j];
        for (int j = 0; j < 3; j++) dfdxi[j] = n[0] * L11[0 + j] + n[1] * L11[3 + j] + n[2] * L11[6 + j];
        for (int j = 0; j < 3; j++) dfdxi[3 + j] = d[0] * B[0 + j] + d[1] * B[3 + j] + d[2] * B[6 + j];
    }
}

/* Triangulator::regDiv, triangulator.cpp:114-128 */
static double tri_reg_div(double num, double denom, double eps)
{
    if (denom > eps * num) return num / denom;
    else if (num == 0) return 2. / eps;
    else return 2. / eps - denom / (num * eps * eps);
}

static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static void mat3_vec(const double A[9], const double v[3], double out[3])
{
    for (int i = 0; i < 3; i++) out[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}

/* Triangulator::computeRegular, triangulator.cpp:145-259.  R, t = transf.rotMat(), transf.trans() (triangulator.h:34-35);
 * any of res1 / res2 / jac1 / jac2 may be NULL (jacN needs resN). */
void vgo_triangulate_regular(const double R[9], const double t[3], double eps, const double p[3], const double q_in[3],
                             double *res1, double *res2, double *jac1, double *jac2)
{
    double q[3], r[3];
    mat3_vec(R, q_in, q); /* q = R * q */
    for (int i = 0; i < 3; i++) r[i] = p[i] + q[i];
    double pq = dot3(p, q);
    double pp = dot3(p, p);
    double qq = dot3(q, q);
    double tp = dot3(t, p);
    double tq = dot3(t, q);
    double tr = dot3(t, r);
    double tt = dot3(t, t);
    double rp = dot3(r, p);
    double rq = dot3(r, q);
    (void)pq; (void)pp; (void)qq;
    double delta = tp * rq - tq * rp;
    double delta1 = 0., delta2 = 0.;
    if (res1) {
        delta1 = tt * rq - tr * tq;
        *res1 = tri_reg_div(delta1, delta, eps);
    }
    if (res2) {
        delta2 = tt * rp - tr * tp;
        *res2 = tri_reg_div(delta2, delta, eps);
    }
    if (jac1 || jac2) {
        double deltaInv = 1. / delta;
        double qSkew[9];
        hat_(q, qSkew);
        double jacDeltaV[3], jacDeltaOmega[3], w[3];
        for (int i = 0; i < 3; i++) jacDeltaV[i] = rq * p[i] - rp * q[i];
        for (int i = 0; i < 3; i++) w[i] = tp * p[i] - tq * p[i] - rp * t[i];
        mat3_vec(qSkew, w, jacDeltaOmega);
        if (jac1) {
            double d1V[3], d1O[3];
            for (int i = 0; i < 3; i++) d1V[i] = 2 * rq * t[i] - tq * r[i] - tr * q[i];
            for (int i = 0; i < 3; i++) w[i] = tt * p[i] - tq * t[i] - tr * t[i];
            mat3_vec(qSkew, w, d1O);
            if (delta > eps * delta1) {
                for (int i = 0; i < 3; i++) jac1[i] = deltaInv * (d1V[i] - *res1 * jacDeltaV[i]);
                for (int i = 0; i < 3; i++) jac1[3 + i] = deltaInv * (d1O[i] - *res1 * jacDeltaOmega[i]);
            } else if (delta1 == 0) {
                for (int i = 0; i < 6; i++) jac1[i] = 0.;
            } else {
                const double coef = 1. / (eps * eps);
                double deltaInv1 = 1. / delta1;
                double k = -coef * deltaInv1;
                double k1 = coef * delta * deltaInv1 * deltaInv1;
                for (int i = 0; i < 3; i++) jac1[i] = k * jacDeltaV[i] + k1 * d1V[i];
                for (int i = 0; i < 3; i++) jac1[3 + i] = k * jacDeltaOmega[i] + k1 * d1O[i];
            }
        }
        if (jac2) {
            double d2V[3], d2O[3];
            for (int i = 0; i < 3; i++) d2V[i] = 2 * rp * t[i] - tp * r[i] - tr * p[i];
            for (int i = 0; i < 3; i++) w[i] = tt * p[i] - tp * t[i];
            mat3_vec(qSkew, w, d2O);
            if (delta > eps * delta2) {
                for (int i = 0; i < 3; i++) jac2[i] = deltaInv * (d2V[i] - *res2 * jacDeltaV[i]);
                for (int i = 0; i < 3; i++) jac2[3 + i] = deltaInv * (d2O[i] - *res2 * jacDeltaOmega[i]);
            } else if (delta2 == 0) {
                for (int i = 0; i < 6; i++) jac2[i] = 0.;
            } else {
                const double coef = -1. / (eps * eps);
                double deltaInv2 = 1. / delta2;
                double k = coef * deltaInv2;
                double k2 = coef * delta * deltaInv2 * deltaInv2;
                for (int i = 0; i < 3; i++) jac2[i] = k * jacDeltaV[i] - k2 * d2V[i];
                for (int i = 0; i < 3; i++) jac2[3 + i] = k * jacDeltaOmega[i] - k2 * d2O[i];
            }
        }
    }
}

/* MonoReprojectCost::Evaluate, local_cost_functions.cpp:216-278.  Blocks [6 (xiOdom), 5 (lengths)], 10 residuals.
 * x1 [5][3] = _xVec1, p2 [5][2] = _pVec2.  jac_odom [10 x 6], jac_len [10 x 5], row-major, either may be NULL. */
void vgo_mono_reproject(int model, const double *intr, const double xiBaseCam[6], const double *x1, const double *p2,
                        const double xiOdom[6], const double lengths[5], double residual[10], double *jac_odom,
                        double *jac_len)
{
    /* :221  xi21 = _xiBaseCam.inverseCompose(xiOdom.inverseCompose(_xiBaseCam)) */
    double inner[6], xi21[6];
    inverse_compose(xiOdom, xiBaseCam, inner);
    inverse_compose(xiBaseCam, inner, xi21);
    /* :224-229  xVec2[i] = _xVec1[i] * params[1][i];  xi21.transform(xVec2, xVec2): R * x, then + trans */
    double R21[9], xv2[5][3];
    vgo_rotation_matrix(xi21 + 3, R21);
    for (int i = 0; i < 5; i++) {
        double s[3] = {x1[3 * i] * lengths[i], x1[3 * i + 1] * lengths[i], x1[3 * i + 2] * lengths[i]}, rx[3];
        mat3_vec(R21, s, rx);
        for (int k = 0; k < 3; k++) xv2[i][k] = rx[k] + xi21[k];
    }
    /* :232-244 */
    for (int i = 0; i < 5; i++) {
        double uv[2];
        if (vgo_project_point(model, intr, xv2[i], uv)) {
            residual[2 * i] = uv[0] - p2[2 * i];
            residual[2 * i + 1] = uv[1] - p2[2 * i + 1];
        } else {
            residual[2 * i + 1] = residual[2 * i] = VGO_DOUBLE_BIG;
        }
    }
    /* :249-259  InterJacobian(_camera, _xiBaseCam.inverse(), xiOdom, JAC_INVERTED) */
    if (jac_odom) {
        double inv[6];
        inverse_(xiBaseCam, inv);
        inter_jacobian ij;
        inter_jacobian_init(&ij, inv, xiOdom, 1);
        for (int i = 0; i < 5; i++) inter_jacobian_dpdxi(&ij, model, intr, xv2[i], jac_odom + i * 12, jac_odom + i * 12 + 6);
    }
    /* :262-275  length jacobian: only (row 2i, col i) and (row 2i + 1, col i) are non-zero */
    if (jac_len) {
        for (int i = 0; i < 50; i++) jac_len[i] = 0;
        for (int i = 0; i < 5; i++) {
            double n2[3], P[6];
            mat3_vec(R21, x1 + 3 * i, n2);
            vgo_projection_jacobian(model, intr, xv2[i], P, P + 3);
            jac_len[i * 11] = P[0] * n2[0] + P[1] * n2[1] + P[2] * n2[2];
            jac_len[i * 11 + 5] = P[3] * n2[0] + P[4] * n2[1] + P[5] * n2[2];
        }
    }
}

/* SparseReprojectCost::Evaluate, local_cost_functions.cpp:281-391.  One block [6 (xiOdom)], 2n residuals.
 * x1, x2 [n][3] = _xVec1, _xVec2 (direction vectors in frames 1 / 2), p2 [n][2], size [n] = _sizeVec.
 * jac [2n x 6] row-major or NULL.  NOTE (:383-389): only the u-row of every point is divided by its size; the v-row
 * keeps the undivided Jacobian although both residuals are divided (:312) -- reproduced as written. */
void vgo_sparse_reproject(int model, const double *intr, const double xiBaseCam[6], int n, const double *x1, const double *x2,
                          const double *p2, const double *size, const double xiOdom[6], double *residual, double *jac)
{
    /* :286  xi12 = _xiBaseCam.inverseCompose(xiOdom.compose(_xiBaseCam)) */
    double inner[6], xi12[6];
    compose_(xiOdom, xiBaseCam, inner);
    inverse_compose(xiBaseCam, inner, xi12);
    /* :291 Triangulator(xi12): R = rotMat(), t = trans(), eps = 1e-3 (triangulator.h:34-35) */
    double Rt[9];
    vgo_rotation_matrix(xi12 + 3, Rt);
    const double eps = 1e-3;
    double R21[9], n12[3] = {-xi12[3], -xi12[4], -xi12[5]};
    vgo_rotation_matrix(n12, R21); /* xi12.rotMatInv(): inverseRotate (:308) and the length Jacobian (:338) */
    double *lam = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *jv = (double *)malloc(sizeof(double) * 6 * (size_t)(n > 0 ? n : 1));
    double *xv2 = (double *)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++)
        vgo_triangulate_regular(Rt, xi12, eps, x1 + 3 * i, x2 + 3 * i, lam + i, NULL, jac ? jv + 6 * i : NULL, NULL);
    /* :303-308  xVec2 = xVec1 * lambda; xi12.inverseTransform: (x - trans), then R(-rot) * */
    for (int i = 0; i < n; i++) {
        double d[3];
        for (int k = 0; k < 3; k++) d[k] = x1[3 * i + k] * lam[i] - xi12[k];
        mat3_vec(R21, d, xv2 + 3 * i);
    }
    /* :311-323 */
    for (int i = 0; i < n; i++) {
        double uv[2];
        if (vgo_project_point(model, intr, xv2 + 3 * i, uv)) {
            residual[2 * i] = (uv[0] - p2[2 * i]) / size[i];
            residual[2 * i + 1] = (uv[1] - p2[2 * i + 1]) / size[i];
        } else {
            residual[2 * i + 1] = residual[2 * i] = VGO_DOUBLE_BIG;
        }
    }
    if (jac) {
        /* :331-345 odometry jacobian */
        double inv[6];
        inverse_(xiBaseCam, inv);
        inter_jacobian ij;
        inter_jacobian_init(&ij, inv, xiOdom, 1);
        for (int i = 0; i < n; i++) {
            if (residual[2 * i] == VGO_DOUBLE_BIG) {
                for (int k = 0; k < 12; k++) jac[i * 12 + k] = 0;
            } else {
                inter_jacobian_dpdxi(&ij, model, intr, xv2 + 3 * i, jac + i * 12, jac + i * 12 + 6);
            }
        }
        /* :348-359 */
        double RcamBase[9], nb[3] = {-xiBaseCam[3], -xiBaseCam[4], -xiBaseCam[5]}, Mo[9], M[9];
        vgo_rotation_matrix(nb, RcamBase);
        vgo_inter_omega_rot(xiOdom + 3, Mo);
        mat3_mul(RcamBase, Mo, M);
        double R21T[9], A[9], tBaseCam1[3], Hn[9], Q[9];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) R21T[3 * r + c] = R21[3 * c + r];
        mat3_mul(RcamBase, R21T, A);       /* (RcamBase * R21.transpose()) */
        mat3_vec(A, xiBaseCam, tBaseCam1); /*   * _xiBaseCam.trans()       */
        hat_(tBaseCam1, Hn);
        for (int k = 0; k < 9; k++) Hn[k] = -Hn[k];
        mat3_mul(Hn, M, Q);                /* Q = -hat(tBaseCam1) * M */
        /* :360-381 */
        for (int i = 0; i < n; i++) {
            if (residual[2 * i] == VGO_DOUBLE_BIG) continue;
            double n2[3], P[6];
            mat3_vec(R21, x1 + 3 * i, n2);
            vgo_projection_jacobian(model, intr, xv2 + 3 * i, P, P + 3);
            const double dpdl0 = P[0] * n2[0] + P[1] * n2[1] + P[2] * n2[2];
            const double dpdl1 = P[3] * n2[0] + P[4] * n2[1] + P[5] * n2[2];
            const double *dldv = jv + 6 * i, *dldw = jv + 6 * i + 3;
            double dldt[3], dldr[3];
            for (int j = 0; j < 3; j++) dldt[j] = dldv[0] * RcamBase[0 + j] + dldv[1] * RcamBase[3 + j] + dldv[2] * RcamBase[6 + j];
            for (int j = 0; j < 3; j++)
                dldr[j] = (dldw[0] * M[0 + j] + dldw[1] * M[3 + j] + dldw[2] * M[6 + j]) +
                          (dldv[0] * Q[0 + j] + dldv[1] * Q[3 + j] + dldv[2] * Q[6 + j]);
            double *J = jac + i * 12;
            for (int j = 0; j < 3; j++) {
                J[j] += dpdl0 * dldt[j];
                J[3 + j] += dpdl0 * dldr[j];
                J[6 + j] += dpdl1 * dldt[j];
                J[9 + j] += dpdl1 * dldr[j];
            }
        }
        /* :383-389  for jptr in [i*12, i*12 + 6): /= size  (the u-row only) */
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 6; k++) jac[i * 12 + k] /= size[i];
    }
    free(lam);
    free(jv);
    free(xv2);
}

/* ------------------------------------------------------------------------------------------
 * Geometric pose initialisation: ICamera::reconstructPoint (eucm.h:85-106, ucm.h:81-103,
 * mei.h:90-112), the 4-corner construction of estimateInitialGrid (src/calibration/unified_calibration.cpp:1066-1135),
 * rotationVector (geometry_core.h:120-124 -> Quaternion(R), quaternion.h:52-59), getInitTransform (:311-348).
 * ---------------------------------------------------------------------------------------- */
int vgo_reconstruct_point(int model, const double *p, const double uv[2], double X[3])
{
    if (model == VGO_MODEL_EUCM) { /* eucm.h:85-106 */
        const double alpha = p[0], beta = p[1], fu = p[2], fv = p[3], u0 = p[4], v0 = p[5];
        double xn = (uv[0] - u0) / fu;
        double yn = (uv[1] - v0) / fv;
        double u2 = xn * xn + yn * yn;
        double gamma = 1. - alpha;
        double num = 1. - u2 * alpha * alpha * beta;
        double det = 1 - (alpha - gamma) * beta * u2;
        if (det < 0) return 0;
        double denom = gamma + alpha * sqrt(det);
        X[0] = xn; X[1] = yn; X[2] = num / denom;
        return 1;
    }
    if (model != VGO_MODEL_UCM && model != VGO_MODEL_MEI) return 0;
    /* ucm.h:81-103 and mei.h:90-112: the same formula (Mei ignores its distortion terms here) */
    const double xi = p[0];
    const double fu = model == VGO_MODEL_UCM ? p[1] : p[6], fv = model == VGO_MODEL_UCM ? p[2] : p[7];
    const double u0 = model == VGO_MODEL_UCM ? p[3] : p[8], v0 = model == VGO_MODEL_UCM ? p[4] : p[9];
    double xn = (uv[0] - u0) / fu;
    double yn = (uv[1] - v0) / fv;
    double u2 = xn * xn + yn * yn;
    double gamma = sqrt(1. + u2 * (1 - xi * xi));
    double etanum = -gamma - xi * u2;
    double etadenom = xi * xi * u2 - 1;
    X[0] = xn; X[1] = yn; X[2] = etadenom / (etadenom + xi * etanum);
    return 1;
}

/* rotationVector(R): Quaternion(const Matrix3 &) quaternion.h:52-59, then toRotationVector */
void vgo_rotation_vector(const double R[9], double rot[3])
{
    double q[4];
    q[3] = sqrt(1.0 + (R[0] + R[4] + R[8])) / 2.0;
    double w4 = (4.0 * q[3]);
    q[0] = (R[7] - R[5]) / w4;
    q[1] = (R[2] - R[6]) / w4;
    q[2] = (R[3] - R[1]) / w4;
    vgo_quat_to_rotvec(q, rot);
}

static void normalize3(double v[3])
{
    double n = norm3(v);
    v[0] /= n; v[1] /= n; v[2] /= n;
}

static double dist3(const double a[3], const double b[3])
{
    double d[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    return norm3(d);
}

/* estimateInitialGrid before its Ceres refinement (:1066-1135).  board4 / corners4: the board points and detected corners
 * at idxUL, idxUR, idxBL, idxBR.  Returns 0 when a corner cannot be reconstructed (the reference ignores that return
 * value and would go on with an uninitialised vector). */
int vgo_initial_grid_pose(int model, const double *intr, const double board4[12], const double corners4[8], double xi[6])
{
    double XUL[3], XUR[3], XBL[3], XBR[3];
    if (!vgo_reconstruct_point(model, intr, corners4 + 0, XUL)) return 0;
    if (!vgo_reconstruct_point(model, intr, corners4 + 2, XUR)) return 0;
    if (!vgo_reconstruct_point(model, intr, corners4 + 4, XBL)) return 0;
    if (!vgo_reconstruct_point(model, intr, corners4 + 6, XBR)) return 0;
    normalize3(XUL);
    normalize3(XUR);
    normalize3(XBR);
    normalize3(XBL);
    const double *bUL = board4, *bUR = board4 + 3, *bBL = board4 + 6, *bBR = board4 + 9;
    double exModelU = dist3(bUR, bUL);
    double exModelB = dist3(bBR, bBL);
    double eyModelL = dist3(bBL, bUL);
    double eyModelR = dist3(bBR, bUR);
    double scaleXU = exModelU / dist3(XUR, XUL);
    double scaleXB = exModelB / dist3(XBR, XBL);
    double scaleYL = eyModelL / dist3(XBL, XUL);
    double scaleYR = eyModelR / dist3(XBR, XUR);
    double pos[3], posx[3], posy[3], ex[3], ey[3], ey2[3], ez[3];
    const double s0 = scaleXU < scaleYL ? scaleXU : scaleYL; /* std::min */
    const double s1 = scaleXU < scaleYR ? scaleXU : scaleYR;
    const double s2 = scaleXB < scaleYL ? scaleXB : scaleYL;
    for (int k = 0; k < 3; k++) {
        pos[k] = XUL[k] * s0;
        posx[k] = XUR[k] * s1;
        posy[k] = XBL[k] * s2;
        ex[k] = posx[k] - pos[k];
        ey[k] = posy[k] - pos[k];
    }
    xi[0] = pos[0]; xi[1] = pos[1]; xi[2] = pos[2];
    normalize3(ex);
    /* ey = (Matrix3d::Identity() - ex * ex.transpose()) * ey : the matrix first, then its product with ey */
    for (int i = 0; i < 3; i++) {
        double m0 = (i == 0 ? 1. : 0.) - ex[i] * ex[0];
        double m1 = (i == 1 ? 1. : 0.) - ex[i] * ex[1];
        double m2 = (i == 2 ? 1. : 0.) - ex[i] * ex[2];
        ey2[i] = m0 * ey[0] + m1 * ey[1] + m2 * ey[2];
    }
    normalize3(ey2);
    ez[0] = ex[1] * ey2[2] - ex[2] * ey2[1]; /* ex.cross(ey) */
    ez[1] = ex[2] * ey2[0] - ex[0] * ey2[2];
    ez[2] = ex[0] * ey2[1] - ex[1] * ey2[0];
    const double R[9] = {ex[0], ey2[0], ez[0], ex[1], ey2[1], ez[1], ex[2], ey2[2], ez[2]}; /* R << ex, ey, ez (columns) */
    vgo_rotation_vector(R, xi + 3);
    return 1;
}

/* getInitTransform (:311-348): peel the chain members before / after the one being initialised off a camera-frame pose.
 * chain [n][6]: current value of every chain member (camera side first); init_index: the member to initialise. */
void vgo_init_transform_range(int n, const int *status, int first_index, int last_index, const double *chain, const double xi_in[6],
                              double out[6])
{
    double xi[6], t[6];
    memcpy(xi, xi_in, sizeof xi);
    for (int i = 0; i < n; i++) {
        if (i == first_index) break; /* name == initName: the first occurrence ends the forward loop (:318) */
        else if (status[i] == VGO_TRANSFORM_DIRECT) inverse_compose(chain + 6 * i, xi, t); /* getTransform(name).inverseCompose(xi) */
        else compose_(chain + 6 * i, xi, t);                                              /* getTransform(name).compose(xi)        */
        memcpy(xi, t, sizeof xi);
    }
    for (int i = n - 1; i >= 0; i--) {
        if (i == last_index) { /* name == initName: the last occurrence ends the backward loop (:330-337) */
            if (status[i] == VGO_TRANSFORM_INVERSE) {
                inverse_(xi, t);
                memcpy(xi, t, sizeof xi);
            }
            break;
        } else if (status[i] == VGO_TRANSFORM_DIRECT) vgo_compose_inverse(xi, chain + 6 * i, t); /* xi.composeInverse(getTransform) */
        else compose_(xi, chain + 6 * i, t);                                                    /* xi.compose(getTransform)        */
        memcpy(xi, t, sizeof xi);
    }
    memcpy(out, xi, sizeof xi);
}

void vgo_init_transform(int n, const int *status, int init_index, const double *chain, const double xi_in[6], double out[6])
{
    vgo_init_transform_range(n, status, init_index, init_index, chain, xi_in, out);
}

int vgo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
