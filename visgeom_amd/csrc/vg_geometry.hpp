// vg_geometry.hpp -- SE(3) primitives of the calibration hot path, device + host.
//
// Semantics (branch thresholds, operation order) follow the reference headers cited on each
// function (paths relative to /root/reference); the code is written for one GPU lane per
// transform chain: plain doubles, no matrix class, trig shared between the three places that
// need the same angle (quaternion, Rodrigues matrix, interaction matrix).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

#define VG_HD __host__ __device__ __forceinline__

namespace vg {

struct Quat {
    double x, y, z, w;
};

// Reciprocal and square root for the fused kernels: the hardware's ~2^-23 estimates (v_rcp_f64 / v_rsq_f64) refined by
// Newton steps to 1-2 ulp, without the scaling, denormal and special-value handling of an IEEE division / sqrt (11 and
// ~15 instructions each, three of them per corner).  Arguments are lengths and depths of points in front of a camera.
__device__ __forceinline__ double rcp_nr(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.), r, r);
    return r;
}

// s = sqrt(x), rs = 1 / sqrt(x) in one go; x is floored at 1e-300 (x == 0: s = 1e-150, rs = 1e150, both finite)
__device__ __forceinline__ void sqrt_rsqrt_nr(double x, double &s, double &rs)
{
    x = fmax(x, 1e-300);
    double y = __builtin_amdgcn_rsq(x);
    y = __builtin_fma(0.5 * y, __builtin_fma(-x * y, y, 1.), y);
    y = __builtin_fma(0.5 * y, __builtin_fma(-x * y, y, 1.), y);
    double t = x * y;
    t = __builtin_fma(0.5 * y, __builtin_fma(-t, t, x), t);
    s = t;
    rs = y;
}

// translation + rotation vector, parameter order [t(3), r(3)]  (geometry/transformation.h:46)
struct Transf {
    double t[3];
    double r[3];
};

VG_HD double norm3(const double *v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

VG_HD void sincos_(double x, double *s, double *c)
{
    // same reduction for both -> one range reduction per angle instead of two
    ::sincos(x, s, c);
}

// sin and cos of one angle for the kernels whose rows never leave the CU (fused Gram, pose refinement): three-part Cody-Waite
// reduction by pi/2 (exact products for |x| < 1e4: rotation angles are a few radians) and the fdlibm kernel polynomials on
// [-pi/4, pi/4]; below 1 ulp, about a third of the instructions of the general-purpose ocml routine and no Payne-Hanek branch
// in the dependent chain at the head of every wave.  Larger arguments take the library routine.
__device__ __forceinline__ void sincos_fast(double x, double *s, double *c)
{
    if (!(fabs(x) < 1.0e4)) {
        ::sincos(x, s, c);
        return;
    }
    const double k = __builtin_rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-k, 1.57079632673412561417e+00, x);   // pi/2, first 33 bits: k * this is exact
    r = __builtin_fma(-k, 6.07710050630396597660e-11, r);          // next 33 bits
    r = __builtin_fma(-k, 2.02226624879595063154e-21, r);          // tail
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    ps = __builtin_fma(r * z, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    pc = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.));
    const int q = (int)k;
    const double ss = (q & 1) ? pc : ps, cs = (q & 1) ? ps : pc;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cs : cs;
}

// geometry/geometry_core.h:24-30
VG_HD double sinc_from(double x, double sinx) { return x == 0. ? 1. : sinx / x; }

// Trig of one rotation vector, evaluated once and shared (every consumer below needs the same
// theta = |v|: Quaternion(v) needs sin/cos(theta/2), rotationMatrix(+-v) sin/cos(theta),
// interOmegaRot(v) sin(theta/2) and sin(theta)).
struct RotTrig {
    double th;      // |v|
    double s, c;    // sin(th), cos(th)
    double sh, ch;  // sin(th/2.), cos(th/2.)
};

VG_HD RotTrig rot_trig(const double *v, bool need_full, bool need_half)
{
    RotTrig q;
    q.th = norm3(v);
    q.s = 0.; q.c = 1.; q.sh = 0.; q.ch = 1.;
    if (need_full && !(q.th < 1e-5)) sincos_(q.th, &q.s, &q.c);
    if (need_half && !(fabs(q.th) < 1e-6)) sincos_(q.th / 2., &q.sh, &q.ch);
    return q;
}

// Quaternion(rot)  geometry/quaternion.h:31-50 ; |theta| < 1e-6 -> (rot/2, 1), not normalised
VG_HD Quat quat_from_rotvec(const double *rot, const RotTrig &g)
{
    Quat q;
    if (fabs(g.th) < 1e-6) {
        q.x = rot[0] / 2.;
        q.y = rot[1] / 2.;
        q.z = rot[2] / 2.;
        q.w = 1.;
    } else {
        const double u0 = rot[0] / g.th, u1 = rot[1] / g.th, u2 = rot[2] / g.th;
        q.x = u0 * g.sh;
        q.y = u1 * g.sh;
        q.z = u2 * g.sh;
        q.w = g.ch;
    }
    return q;
}

// geometry/geometry_core.h:32-38
VG_HD double normalize_angle(double th)
{
    if (th > M_PI) return th - 2 * M_PI;
    else if (th < -M_PI) return th + 2 * M_PI;
    else return th;
}

// Quaternion::toRotationVector  geometry/quaternion.h:84-98
VG_HD void quat_to_rotvec(const Quat &q, double *rot)
{
    const double s = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    if (s < 1e-5) {
        rot[0] = q.x * 2.;
        rot[1] = q.y * 2.;
        rot[2] = q.z * 2.;
    } else {
        const double th = 2. * atan2(s, q.w);
        const double thn = normalize_angle(th);
        rot[0] = q.x / s * thn;
        rot[1] = q.y / s * thn;
        rot[2] = q.z / s * thn;
    }
}

// Quaternion::rotate  geometry/quaternion.h:61-82
VG_HD void quat_rotate(const Quat &q, const double *v, double *out)
{
    const double t1 = q.w * q.x;
    const double t2 = q.w * q.y;
    const double t3 = q.w * q.z;
    const double t4 = -q.x * q.x;
    const double t5 = q.x * q.y;
    const double t6 = q.x * q.z;
    const double t7 = -q.y * q.y;
    const double t8 = q.y * q.z;
    const double t9 = -q.z * q.z;
    const double v1 = v[0], v2 = v[1], v3 = v[2];
    out[0] = 2. * ((t7 + t9) * v1 + (t5 - t3) * v2 + (t2 + t6) * v3) + v1;
    out[1] = 2. * ((t3 + t5) * v1 + (t4 + t9) * v2 + (t8 - t1) * v3) + v2;
    out[2] = 2. * ((t6 - t2) * v1 + (t1 + t8) * v2 + (t4 + t7) * v3) + v3;
}

// Quaternion::operator*  geometry/quaternion.h:105-118
VG_HD Quat quat_mul(const Quat &a, const Quat &b)
{
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}

// rotationMatrix(sign * v)  geometry/geometry_core.h:40-76 ; R row-major.
// sign = -1 gives Transformation::rotMatInv() (geometry/transformation.h:132): |-v| == |v| bit for bit.
VG_HD void rotation_matrix(const double *v, double sign, const RotTrig &g, double *R)
{
    const double v0 = sign * v[0], v1 = sign * v[1], v2 = sign * v[2];
    if (g.th < 1e-5) {
        R[0] = 1.;  R[1] = -v2; R[2] = v1;
        R[3] = v2;  R[4] = 1.;  R[5] = -v0;
        R[6] = -v1; R[7] = v0;  R[8] = 1.;
    } else {
        const double thInv = 1. / g.th;
        const double u1 = v0 * thInv;
        const double u2 = v1 * thInv;
        const double u3 = v2 * thInv;
        const double sinth = g.s;
        const double costhVar = 1. - g.c;

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

// C = A * B, 3x3 row-major, each coefficient summed k = 0,1,2
VG_HD void mat3_mul(const double *A, const double *B, double *C)
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            C[3 * i + j] = A[3 * i + 0] * B[0 + j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// interOmegaRot(v)  geometry/geometry_core.h:158-180  (omega = M(r) * rdot)
VG_HD void inter_omega_rot(const double *v, const RotTrig &g, double *B)
{
    if (g.th < 1e-5) {
        const double h0 = v[0] / 2., h1 = v[1] / 2., h2 = v[2] / 2.;
        B[0] = 1.;  B[1] = -h2; B[2] = h1;
        B[3] = h2;  B[4] = 1.;  B[5] = -h0;
        B[6] = -h1; B[7] = h0;  B[8] = 1.;
    } else {
        const double u0 = v[0] / g.th, u1 = v[1] / g.th, u2 = v[2] / g.th;
        // uhat = hat(v / theta)   geometry_core.h:126-132
        const double uhat[9] = {0, -u2, u1, u2, 0, -u0, -u1, u0, 0};
        const double thetaHalf = g.th / 2.;
        double K1 = sinc_from(thetaHalf, g.sh);
        K1 = thetaHalf * K1 * K1;
        const double K2 = (1. - sinc_from(g.th, g.s));
        double k2u[9], prod[9];
#pragma unroll
        for (int i = 0; i < 9; i++) k2u[i] = K2 * uhat[i];
        mat3_mul(k2u, uhat, prod);  // (K2*uhat)*uhat
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double id = (i == j) ? 1. : 0.;
                B[3 * i + j] = (id + K1 * uhat[3 * i + j]) + prod[3 * i + j];
            }
    }
}

// ------------------------------------------------------------------------------------------
// Per-block frame: everything GenericProjectionJac::Evaluate derives from the chain parameters
// and then re-uses for every corner of the image.
//   [0..8]  Racc   = R(xiAcc.rot)                     calib_cost_functions.cpp:49-50
//   [9..11] tacc   = xiAcc.trans
//   member l at 12 + 21*l:  R12[9], M12[9], t13[3]    InterJacobian ctor, jacobian.h:139-152
// Frame stride is padded to an even number of doubles so frames stay 16-byte aligned.
// ------------------------------------------------------------------------------------------
VG_HD constexpr int frame_doubles(int L) { return 12 + 21 * L; }
VG_HD constexpr int frame_stride(int L) { return (frame_doubles(L) + 1) & ~1; }

// The chain walk, one member at a time.  Mirrors calib_cost_functions.cpp:32-46 (accumulation) and :76-92 (xi13 / xi23
// pick); the reference walks the chain twice with identical arithmetic, one walk is enough.
struct ChainState {
    Transf acc;         // xiAcc (acc.r is maintained only while !fast)
    double Racc[9];     // R(xiAcc.rot) when racc_valid
    bool racc_valid;
    Quat qacc;          // fast: xiAcc's rotation as the unit quaternion with w >= 0, i.e. Quaternion(xiAcc.rot)
    bool fast;
};

// The device walk keeps the accumulated rotation as a quaternion wherever that is the reference's arithmetic up to rounding.
// The reference turns every product q1 * q2 back into a rotation vector (atan2), and the next consumer turns that vector into
// a quaternion (sincos of theta/2) or a matrix (sincos of theta) again: Quaternion(toRotationVector(q)) is q / |q| with
// w >= 0, and rotationMatrix(toRotationVector(q)) is the matrix of that unit quaternion, both to 1e-16 -- EXCEPT inside the
// first-order branches (|q.xyz| < 1e-5 in toRotationVector, theta < 1e-5 / 1e-6 in rotationMatrix / Quaternion), whose
// results are up to 5e-11 away from the exact ones.  So the quaternion is carried only while |q.xyz| >= 1e-4 |q| (theta >=
// 2e-4: twenty times the widest threshold); below, the reference-order routines run unchanged.  A [global INVERSE, pose
// DIRECT] chain drops from seven sincos, two atan2 and their square roots / divisions in one dependent chain to two sincos.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VG_WALK_REFERENCE_ORDER)
#define VG_WALK_FAST 1
#else
#define VG_WALK_FAST 0
#endif

// rotation matrix of a unit quaternion, row-major
VG_HD void quat_matrix(const Quat &q, double *R)
{
    const double xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
    const double xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
    const double wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
    R[0] = 1. - 2. * (yy + zz); R[1] = 2. * (xy - wz);      R[2] = 2. * (xz + wy);
    R[3] = 2. * (xy + wz);      R[4] = 1. - 2. * (xx + zz); R[5] = 2. * (yz - wx);
    R[6] = 2. * (xz - wy);      R[7] = 2. * (yz + wx);      R[8] = 1. - 2. * (xx + yy);
}

#if VG_WALK_FAST
// qres = q1 * q2 of a compose / composeInverse: true when its rotation is carried as s.qacc from here on
__device__ __forceinline__ bool chain_carry_quat(ChainState &s, const Quat &qres)
{
    const double v2 = qres.x * qres.x + qres.y * qres.y + qres.z * qres.z;
    const double n2 = v2 + qres.w * qres.w;
    s.fast = v2 >= 1e-8 * n2;
    if (!s.fast) return false;
    double nrm, inv;
    sqrt_rsqrt_nr(n2, nrm, inv);
    if (qres.w < 0.) inv = -inv;   // toRotationVector wraps theta > pi to theta - 2 pi: the quaternion of that vector is -q
    s.qacc = {qres.x * inv, qres.y * inv, qres.z * inv, qres.w * inv};
    return true;
}
#endif

// R(xiAcc.rot) into s.Racc
VG_HD void chain_acc_matrix(ChainState &s)
{
#if VG_WALK_FAST
    if (s.fast) {
        quat_matrix(s.qacc, s.Racc);
        return;
    }
#endif
    const RotTrig go = rot_trig(s.acc.r, true, false);
    rotation_matrix(s.acc.r, 1., go, s.Racc);
}

// trig of a chain member's own rotation vector.  Device: ONE sincos (half angle; sin th = 2 sh ch, cos th = 1 - 2 sh^2, so the
// 1 - cos th of rotationMatrix is 2 sh^2 to an ulp), in exactly the branches rot_trig takes.
VG_HD RotTrig rot_trig_member(const double *v)
{
#if VG_WALK_FAST
    RotTrig q;
    q.th = norm3(v);
    q.s = 0.; q.c = 1.; q.sh = 0.; q.ch = 1.;
    if (!(fabs(q.th) < 1e-6)) sincos_(q.th / 2., &q.sh, &q.ch);
    if (!(q.th < 1e-5)) {
        q.s = 2. * q.sh * q.ch;
        q.c = 1. - 2. * q.sh * q.sh;
    }
    return q;
#else
    return rot_trig(v, true, true);
#endif
}

VG_HD void chain_state_init(ChainState &s)
{
    s.acc = {{0., 0., 0.}, {0., 0., 0.}};  // Transformation() = zeros  transformation.h:36
    const double I0[9] = {1., -0., 0., 0., 1., -0., -0., 0., 1.};  // rotationMatrix(0) = I + hat(0)
#pragma unroll
    for (int i = 0; i < 9; i++) s.Racc[i] = I0[i];
    s.racc_valid = true;
    s.qacc = {0., 0., 0., 1.};
    s.fast = false;   // the identity is inside the first-order branches: reference order
}

// Quaternion(xiAcc.rot): what every member's compose / composeInverse starts with (transformation.h:83,105)
VG_HD Quat chain_acc_quat(const ChainState &s)
{
#if VG_WALK_FAST
    if (s.fast) return s.qacc;
#endif
    const RotTrig gacc = rot_trig(s.acc.r, false, true);
    return quat_from_rotvec(s.acc.r, gacc);
}

// one member: xi23 with its status, q1 = Quaternion(xiAcc.rot) of the state BEFORE it; out = [R12 (9) | M12 (9) | t13 (3)]
VG_HD void chain_walk_member(ChainState &s, const Quat &q1, const double *xi23, bool inverted, double *out)
{
    Transf &acc = s.acc;
    const double t23[3] = {xi23[0], xi23[1], xi23[2]};
    const double r23[3] = {xi23[3], xi23[4], xi23[5]};
    const RotTrig g23 = rot_trig_member(r23);
    const Quat q2 = quat_from_rotvec(r23, g23);

    double R13[9], t13[3];
    if (!inverted) {
        // xiAcc = xiAcc.compose(xi23); xi13 = xiAcc      transformation.h:80-88
        double rt[3];
        quat_rotate(q1, t23, rt);
        const Quat qres = quat_mul(q1, q2);
        acc.t[0] = rt[0] + acc.t[0];
        acc.t[1] = rt[1] + acc.t[1];
        acc.t[2] = rt[2] + acc.t[2];
#if VG_WALK_FAST
        if (!chain_carry_quat(s, qres))
#endif
            quat_to_rotvec(qres, acc.r);
        chain_acc_matrix(s);
        s.racc_valid = true;
#pragma unroll
        for (int i = 0; i < 9; i++) R13[i] = s.Racc[i];
        t13[0] = acc.t[0]; t13[1] = acc.t[1]; t13[2] = acc.t[2];
    } else {
        // xi13 = xiAcc; xiAcc = xiAcc.composeInverse(xi23)   transformation.h:101-110
        if (!s.racc_valid) chain_acc_matrix(s);
#pragma unroll
        for (int i = 0; i < 9; i++) R13[i] = s.Racc[i];
        t13[0] = acc.t[0]; t13[1] = acc.t[1]; t13[2] = acc.t[2];
        const Quat q2inv = {-q2.x, -q2.y, -q2.z, q2.w};  // quaternion.h:100-103
        const Quat qres = quat_mul(q1, q2inv);
        double rt[3];
        quat_rotate(qres, t23, rt);
        acc.t[0] = acc.t[0] - rt[0];
        acc.t[1] = acc.t[1] - rt[1];
        acc.t[2] = acc.t[2] - rt[2];
#if VG_WALK_FAST
        if (!chain_carry_quat(s, qres))
#endif
            quat_to_rotvec(qres, acc.r);
        s.racc_valid = false;
    }

    // InterJacobian(camera, xi13, xi23, inverted)   jacobian.h:139-152
    double Rb[9], M[9], R12[9], M12[9];
    rotation_matrix(r23, -1., g23, Rb);  // xi23.rotMatInv()
    mat3_mul(R13, Rb, R12);
    inter_omega_rot(r23, g23, M);
    mat3_mul(R12, M, M12);
#pragma unroll
    for (int i = 0; i < 9; i++) out[i] = inverted ? R12[i] * -1 : R12[i];
#pragma unroll
    for (int i = 0; i < 9; i++) out[9 + i] = inverted ? M12[i] * -1 : M12[i];
    out[18] = t13[0]; out[19] = t13[1]; out[20] = t13[2];
}

// frame[0..11] = R(xiAcc.rot), xiAcc.trans of the finished chain
VG_HD void chain_finish(ChainState &s, double *frame)
{
    if (!s.racc_valid) chain_acc_matrix(s);
#pragma unroll
    for (int i = 0; i < 9; i++) frame[i] = s.Racc[i];
    frame[9] = s.acc.t[0]; frame[10] = s.acc.t[1]; frame[11] = s.acc.t[2];
}

// One chain walk producing the frame.  `member(l)` returns a pointer to the 6-vector of chain member l.
template <typename MemberFn>
VG_HD void build_frame(int L, const int *status, MemberFn member, double *frame)
{
    ChainState s;
    chain_state_init(s);
    for (int l = 0; l < L; l++) {
        const Quat q1 = chain_acc_quat(s);
        chain_walk_member(s, q1, member(l), status[l] != 0, frame + 12 + 21 * l);
    }
    chain_finish(s, frame);
}

// The same walk for members that do not lie in one parameter vector (the candidate point of an LM step inside the
// back-substitution kernel: pose from registers, global members from LDS, constant members from memory): `fill(l, xi)`
// writes the 6-vector of member l.  Same arithmetic in the same order as build_frame(): same bits.
template <typename FillFn>
VG_HD void build_frame_vals(int L, const int *status, FillFn fill, double *frame)
{
    ChainState s;
    chain_state_init(s);
    for (int l = 0; l < L; l++) {
        const Quat q1 = chain_acc_quat(s);
        double xi[6];
        fill(l, xi);
        chain_walk_member(s, q1, xi, status[l] != 0, frame + 12 + 21 * l);
    }
    chain_finish(s, frame);
}

// The frame of a chain with ONE member used DIRECT (the mono calibration case): xiAcc = identity o xi23 = xi23.
// build_frame() gets there through the reference's rotvec -> quaternion -> rotvec round trip and a second Rodrigues
// evaluation; that round trip is the identity up to rounding (|dR| < 1e-15, also across its first-order branches,
// where the rotation vector moves by theta^3 / 24 < 4e-16), so this routine evaluates the trig once and is short enough
// (about 300 instructions) to run inside the emit kernel instead of as a launch of its own.
VG_HD void build_frame_single_direct(const double *xi, double *frame)
{
    const double r[3] = {xi[3], xi[4], xi[5]};
    const RotTrig g = rot_trig(r, true, true);
    // R13 = R(xiAcc.rot).  The reference's xiAcc.rot is not xi.rot itself but its image under compose()'s round trip
    // through the quaternion: below |q.xyz| = 1e-5 that is 2 sin(th/2) u (quaternion.h:88-91), SHORTER than th by
    // th^3/24.  Nothing but rounding for the values -- but just above th = 1e-5 it carries xiAcc.rot back under the
    // first-order threshold of rotationMatrix (geometry_core.h:45), so the reference takes the first-order R13 there
    // while th itself says Rodrigues (5e-11 apart: above the 1e-10 bar once projected).  In that band the round trip
    // is evaluated exactly as the reference does and decides the branch.
    double racc[3] = {r[0], r[1], r[2]};
    RotTrig gacc = g;
    if (!(g.th < 1e-5) && g.th < 1.00000001e-5) {
        // Quaternion(rot) (quaternion.h:41-48), identity * q == q component by component, toRotationVector (:86-91):
        // here |q.xyz| = sin(th / 2) ~ 5e-6 is always below its 1e-5 threshold, so the atan2 branch cannot occur
        // (written out instead of calling quat_to_rotvec: its unreachable branch cost the kernel 3 % in registers)
        const double x = r[0] / g.th * g.sh, y = r[1] / g.th * g.sh, z = r[2] / g.th * g.sh;
        if (sqrt(x * x + y * y + z * z) < 1e-5) {
            racc[0] = x * 2.;
            racc[1] = y * 2.;
            racc[2] = z * 2.;
            gacc.th = norm3(racc);  // sin / cos of it: those of th to 1e-16 (only used if still >= 1e-5)
        }
    }
    double R[9], Rb[9], M[9], R12[9], M12[9];
    rotation_matrix(racc, 1., gacc, R);  // R13 = R(xiAcc.rot)
    rotation_matrix(r, -1., g, Rb);   // xi23.rotMatInv()
    mat3_mul(R, Rb, R12);             // jacobian.h:142 (the identity up to rounding, kept as computed)
    inter_omega_rot(r, g, M);
    mat3_mul(R12, M, M12);
#pragma unroll
    for (int i = 0; i < 9; i++) {
        frame[i] = R[i];
        frame[12 + i] = R12[i];
        frame[21 + i] = M12[i];
    }
    frame[9] = xi[0]; frame[10] = xi[1]; frame[11] = xi[2];
    frame[30] = xi[0]; frame[31] = xi[1]; frame[32] = xi[2];
}

// The same frame for consumers that only need it to rounding-level agreement (the fused Gram kernels, whose rows never
// leave the CU and are held to 1e-10 of the reference-order Gram): ONE sincos (half angle; sin th = 2 sh ch,
// 1 - cos th = 2 sh^2), R12 = R(r) R(-r) taken as the identity it is to 1e-16, M from uhat^2 = u u^T - I.  About a third
// of the dependent instruction chain of build_frame_single_direct -- the chain walk is a serial prologue of every
// workgroup of vg_gram_valu_kernel.  Below the reference's first-order threshold (theta < 1e-5, where its R12 is
// I - hat(r)^2, 1e-10 away from I) and in the band just above it the reference-order routine is used unchanged.
__device__ __forceinline__ void build_frame_single_direct_fast(const double *xi, double *frame)
{
#pragma clang fp contract(fast)
    const double r0 = xi[3], r1 = xi[4], r2 = xi[5];
    double th, ti;
    sqrt_rsqrt_nr(r0 * r0 + r1 * r1 + r2 * r2, th, ti);
    if (th < 1.00000001e-5) {  // first-order forms, and the band where the reference's round trip decides the branch
        build_frame_single_direct(xi, frame);
        return;
    }
    const double u0 = r0 * ti, u1 = r1 * ti, u2 = r2 * ti;
    const double h = 0.5 * th;
    double sh, ch;
    sincos_fast(h, &sh, &ch);
    const double s = 2. * sh * ch, cv = 2. * sh * sh;  // sin(theta), 1 - cos(theta)
    // Rodrigues, geometry_core.h:53-75
    frame[0] = 1. + cv * (u0 * u0 - 1.);
    frame[4] = 1. + cv * (u1 * u1 - 1.);
    frame[8] = 1. + cv * (u2 * u2 - 1.);
    frame[1] = -s * u2 + cv * u0 * u1;
    frame[2] = s * u1 + cv * u0 * u2;
    frame[5] = -s * u0 + cv * u1 * u2;
    frame[3] = s * u2 + cv * u1 * u0;
    frame[6] = -s * u1 + cv * u2 * u0;
    frame[7] = s * u0 + cv * u2 * u1;
    // interOmegaRot, geometry_core.h:170-179: I + K1 uhat + K2 uhat^2,  K1 = h sinc(h)^2,  K2 = 1 - sinc(theta)
    const double K1 = sh * sh * (2. * ti), K2 = 1. - s * ti;  // sh^2 / h with 1 / h = 2 / theta
    frame[21] = 1. + K2 * (u0 * u0 - 1.);
    frame[25] = 1. + K2 * (u1 * u1 - 1.);
    frame[29] = 1. + K2 * (u2 * u2 - 1.);
    frame[22] = -K1 * u2 + K2 * u0 * u1;
    frame[23] = K1 * u1 + K2 * u0 * u2;
    frame[26] = -K1 * u0 + K2 * u1 * u2;
    frame[24] = K1 * u2 + K2 * u1 * u0;
    frame[27] = -K1 * u1 + K2 * u2 * u0;
    frame[28] = K1 * u0 + K2 * u2 * u1;
#pragma unroll
    for (int i = 0; i < 9; i++) frame[12 + i] = (i % 4 == 0) ? 1. : 0.;
    frame[9] = xi[0]; frame[10] = xi[1]; frame[11] = xi[2];
    frame[30] = xi[0]; frame[31] = xi[1]; frame[32] = xi[2];
}

}  // namespace vg
