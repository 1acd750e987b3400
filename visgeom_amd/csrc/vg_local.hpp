// vg_local.hpp -- SURVEY 8(f) rank 5: the localization reprojection costs on the device camera models.
//
//   MonoReprojectCost::Evaluate    src/localization/local_cost_functions.cpp:216-278   blocks [6, 5], 10 residuals
//   SparseReprojectCost::Evaluate  src/localization/local_cost_functions.cpp:281-391   block [6], 2 n residuals
//   Triangulator::computeRegular   src/reconstruction/triangulator.cpp:145-259 (+ regDiv :114-128)
//   CameraJacobian                 include/projection/jacobian.h:51-119
//
// In the reference these run one block at a time inside ceres::Solve; SparseOdometry::ransacNPoints
// (src/localization/sparse_odom.cpp:511-606) solves 200 independent few-point problems per frame pair and then one
// problem on all inliers.  Here a SET of blocks is resident in HBM and evaluated per launch: the block frames (the
// transform chain, the InterJacobian constructor, the matrices of the depth Jacobian: one lane per block) and the points (one
// lane per feature: triangulate, transform, project, 2 x 6 rows), rows streamed out through the same wave tile as the
// calibration emit kernel.  Five-point blocks (mono) and small sparse launches compute their frames inside the point kernel
// (LDS); large sparse sets keep a frame launch of their own (hundreds of points share a frame).  Arithmetic in the reference's order (-ffp-contract=off); parity tests:
// tests/test_gpu_local_costs.py (<= 1e-10).
#pragma once

#include "vg_kernels.hpp"

namespace vg {

// ---- Transformation<double> pieces on raw 6-vectors [t, r] (include/geometry/transformation.h) ----
VG_HD Quat quat_of_rotvec(const double *rot)
{
    const RotTrig g = rot_trig(rot, false, true);
    return quat_from_rotvec(rot, g);
}

// compose  transformation.h:80-88
VG_HD void transf_compose(const double *a, const double *b, double *out)
{
    const Quat q1 = quat_of_rotvec(a + 3), q2 = quat_of_rotvec(b + 3);
    double rt[3];
    quat_rotate(q1, b, rt);
    out[0] = rt[0] + a[0];
    out[1] = rt[1] + a[1];
    out[2] = rt[2] + a[2];
    quat_to_rotvec(quat_mul(q1, q2), out + 3);
}

// inverseCompose  transformation.h:90-99  (a^-1 o b)
VG_HD void transf_inverse_compose(const double *a, const double *b, double *out)
{
    const Quat q1 = quat_of_rotvec(a + 3), q2 = quat_of_rotvec(b + 3);
    const Quat q1inv = {-q1.x, -q1.y, -q1.z, q1.w};
    const double d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    quat_rotate(q1inv, d, out);
    quat_to_rotvec(quat_mul(q1inv, q2), out + 3);
}

// the same two compositions with the quaternions of the rotation parts given (computed by quat_of_rotvec on the same device:
// the same bits as the functions above produce inside)
VG_HD void transf_compose_q(const double *a, const Quat &q1, const double *b, const Quat &q2, double *out)
{
    double rt[3];
    quat_rotate(q1, b, rt);
    out[0] = rt[0] + a[0];
    out[1] = rt[1] + a[1];
    out[2] = rt[2] + a[2];
    quat_to_rotvec(quat_mul(q1, q2), out + 3);
}

VG_HD void transf_inverse_compose_q(const double *a, const Quat &q1, const double *b, const Quat &q2, double *out)
{
    const Quat q1inv = {-q1.x, -q1.y, -q1.z, q1.w};
    const double d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    quat_rotate(q1inv, d, out);
    quat_to_rotvec(quat_mul(q1inv, q2), out + 3);
}

// rotMat() / rotMatInv()  transformation.h:131-132
VG_HD void transf_rot_mat(const double *a, double sign, double *R)
{
    const RotTrig g = rot_trig(a + 3, true, false);
    rotation_matrix(a + 3, sign, g, R);
}

// inverse  transformation.h:112-119 : t = -R(-r) t, r = -r
VG_HD void transf_inverse(const double *a, double *out)
{
    double R[9];
    transf_rot_mat(a, -1., R);
#pragma unroll
    for (int i = 0; i < 3; i++) out[i] = (-R[3 * i]) * a[0] + (-R[3 * i + 1]) * a[1] + (-R[3 * i + 2]) * a[2];
    out[3] = -a[3];
    out[4] = -a[4];
    out[5] = -a[5];
}

VG_HD void mat3_vec(const double *A, const double *v, double *out)
{
#pragma unroll
    for (int i = 0; i < 3; i++) out[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}

// InterJacobian ctor  include/projection/jacobian.h:139-152 -> fm = [R12 (9) | M12 (9) | t13 (3)], what pose_rows reads
VG_HD void inter_jacobian_frame(const double *xi13, const double *xi23, bool inverted, double *fm)
{
    double Ra[9], Rb[9], M[9], R12[9], M12[9];
    transf_rot_mat(xi13, 1., Ra);
    const RotTrig g23 = rot_trig(xi23 + 3, true, true);
    rotation_matrix(xi23 + 3, -1., g23, Rb);
    mat3_mul(Ra, Rb, R12);
    inter_omega_rot(xi23 + 3, g23, M);
    mat3_mul(R12, M, M12);
#pragma unroll
    for (int i = 0; i < 9; i++) {
        fm[i] = inverted ? R12[i] * -1 : R12[i];
        fm[9 + i] = inverted ? M12[i] * -1 : M12[i];
    }
    fm[18] = xi13[0];
    fm[19] = xi13[1];
    fm[20] = xi13[2];
}

// ------------------------------------------------------------------------------------------ per-block frames
// mono:   [t21 (3) | R21 = rotMat(xi21) (9) | InterJacobian (21)]                                          = 33 doubles
// sparse: [t12 (3) | Rt = rotMat(xi12) (9) | R21 = rotMatInv(xi12) (9) | InterJacobian (21) | RcamBase (9) | M (9) | Q (9)] = 69
constexpr int kMonoFrame = 34;    // padded to even: 16-byte aligned frames
constexpr int kSparseFrame = 70;
constexpr int kMonoPoints = 5;    // "5-point algorithm", local_cost_functions.h:156-168

// ---- what the block frames need of xiBaseCam ALONE -- a constructor argument of the cost functions (local_cost_functions.h:
// 163, 189), the same for every block of a set and for every evaluation: Quaternion(xiBaseCam.rot) (used by both compositions
// of the chain), xiBaseCam.inverse() and its rotation matrix (the InterJacobian's xi13), xiBaseCam.rotMatInv() (the depth
// Jacobian).  Computed ONCE per set by vg_local_base_kernel -- on the device, by the functions the frames used to call, so the
// frames keep their bits -- instead of by every block's lane in every evaluation: a third of the dependent chain that IS the
// run time of the frame phase.   [qb (4) | inv (6) | R(inv) (9) | RcamBase (9)]
constexpr int kBaseConst = 28;

VG_HD void base_const(const double *xiBaseCam, double *c)
{
    const Quat qb = quat_of_rotvec(xiBaseCam + 3);
    c[0] = qb.x; c[1] = qb.y; c[2] = qb.z; c[3] = qb.w;
    transf_inverse(xiBaseCam, c + 4);
    transf_rot_mat(c + 4, 1., c + 10);
    transf_rot_mat(xiBaseCam, -1., c + 19);
}

// inter_jacobian_frame with R(xi13) and the trigonometry of xi23's rotation given (every consumer of one rotation vector shares
// ONE evaluation of its norm, sine and cosine: the same values the separate calls produced)
VG_HD void inter_jacobian_frame_r(const double *xi13, const double *Ra, const double *xi23, const RotTrig &g23, bool inverted, double *fm)
{
    double Rb[9], M[9], R12[9], M12[9];
    rotation_matrix(xi23 + 3, -1., g23, Rb);
    mat3_mul(Ra, Rb, R12);
    inter_omega_rot(xi23 + 3, g23, M);
    mat3_mul(R12, M, M12);
#pragma unroll
    for (int i = 0; i < 9; i++) {
        fm[i] = inverted ? R12[i] * -1 : R12[i];
        fm[9 + i] = inverted ? M12[i] * -1 : M12[i];
    }
    fm[18] = xi13[0];
    fm[19] = xi13[1];
    fm[20] = xi13[2];
}

// mono_frame / sparse_frame (below) from the set's base constants
VG_HD void mono_frame_c(const double *xiBaseCam, const double *c, const double *xiOdom, double *f)
{
    const Quat qb = {c[0], c[1], c[2], c[3]};
    double inner[6], xi21[6];
    const RotTrig go = rot_trig(xiOdom + 3, true, true);
    transf_inverse_compose_q(xiOdom, quat_from_rotvec(xiOdom + 3, go), xiBaseCam, qb, inner);
    transf_inverse_compose_q(xiBaseCam, qb, inner, quat_of_rotvec(inner + 3), xi21);
    f[0] = xi21[0];
    f[1] = xi21[1];
    f[2] = xi21[2];
    transf_rot_mat(xi21, 1., f + 3);
    inter_jacobian_frame_r(c + 4, c + 10, xiOdom, go, true, f + 12);
}

// local_cost_functions.cpp:219-221 (chain), :251-252 (InterJacobian)
VG_HD void mono_frame(const double *xiBaseCam, const double *xiOdom, double *f)
{
    double inner[6], xi21[6], inv[6];
    transf_inverse_compose(xiOdom, xiBaseCam, inner);
    transf_inverse_compose(xiBaseCam, inner, xi21);     // xi21 = _xiBaseCam.inverseCompose(xiOdom.inverseCompose(_xiBaseCam))
    f[0] = xi21[0];
    f[1] = xi21[1];
    f[2] = xi21[2];
    transf_rot_mat(xi21, 1., f + 3);                    // xi21.rotMat(): transform (:229) and the length Jacobian (:264)
    transf_inverse(xiBaseCam, inv);
    inter_jacobian_frame(inv, xiOdom, true, f + 12);    // InterJacobian(_camera, _xiBaseCam.inverse(), xiOdom, JAC_INVERTED)
}

// local_cost_functions.cpp:285-286 (chain), :291 (Triangulator), :331-332 (InterJacobian), :347-359 (depth Jacobian)
VG_HD void sparse_frame(const double *xiBaseCam, const double *xiOdom, double *f)
{
    double inner[6], xi12[6], inv[6];
    transf_compose(xiOdom, xiBaseCam, inner);
    transf_inverse_compose(xiBaseCam, inner, xi12);     // xi12 = _xiBaseCam.inverseCompose(xiOdom.compose(_xiBaseCam))
    f[0] = xi12[0];
    f[1] = xi12[1];
    f[2] = xi12[2];
    double *Rt = f + 3, *R21 = f + 12, *RcamBase = f + 42, *M = f + 51, *Q = f + 60;
    transf_rot_mat(xi12, 1., Rt);                       // Triangulator: R(transf.rotMat()), t(transf.trans())
    transf_rot_mat(xi12, -1., R21);                     // xi12.rotMatInv()
    transf_inverse(xiBaseCam, inv);
    inter_jacobian_frame(inv, xiOdom, true, f + 21);
    transf_rot_mat(xiBaseCam, -1., RcamBase);           // _xiBaseCam.rotMatInv()
    double Mo[9], R21T[9], A[9], tBaseCam1[3];
    const RotTrig go = rot_trig(xiOdom + 3, true, true);
    inter_omega_rot(xiOdom + 3, go, Mo);
    mat3_mul(RcamBase, Mo, M);                          // M = RcamBase * interOmegaRot(xiOdom.rot())
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) R21T[3 * r + c] = R21[3 * c + r];
    mat3_mul(RcamBase, R21T, A);
    mat3_vec(A, xiBaseCam, tBaseCam1);                  // tBaseCam1 = RcamBase * R21.transpose() * _xiBaseCam.trans()
    const double Hn[9] = {-0., tBaseCam1[2], -tBaseCam1[1], -tBaseCam1[2], -0., tBaseCam1[0], tBaseCam1[1], -tBaseCam1[0], -0.};
    mat3_mul(Hn, M, Q);                                 // Q = -hat(tBaseCam1) * M
}

VG_HD void sparse_frame_c(const double *xiBaseCam, const double *c, const double *xiOdom, double *f)
{
    const Quat qb = {c[0], c[1], c[2], c[3]};
    double inner[6], xi12[6];
    const RotTrig go = rot_trig(xiOdom + 3, true, true);
    transf_compose_q(xiOdom, quat_from_rotvec(xiOdom + 3, go), xiBaseCam, qb, inner);
    transf_inverse_compose_q(xiBaseCam, qb, inner, quat_of_rotvec(inner + 3), xi12);
    f[0] = xi12[0];
    f[1] = xi12[1];
    f[2] = xi12[2];
    double *Rt = f + 3, *R21 = f + 12, *RcamBase = f + 42, *M = f + 51, *Q = f + 60;
    const RotTrig g12 = rot_trig(xi12 + 3, true, false);
    rotation_matrix(xi12 + 3, 1., g12, Rt);
    rotation_matrix(xi12 + 3, -1., g12, R21);
    inter_jacobian_frame_r(c + 4, c + 10, xiOdom, go, true, f + 21);
#pragma unroll
    for (int i = 0; i < 9; i++) RcamBase[i] = c[19 + i];
    double Mo[9], R21T[9], A[9], tBaseCam1[3];
    inter_omega_rot(xiOdom + 3, go, Mo);
    mat3_mul(RcamBase, Mo, M);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) R21T[3 * r + cc] = R21[3 * cc + r];
    mat3_mul(RcamBase, R21T, A);
    mat3_vec(A, xiBaseCam, tBaseCam1);
    const double Hn[9] = {-0., tBaseCam1[2], -tBaseCam1[1], -tBaseCam1[2], -0., tBaseCam1[0], tBaseCam1[1], -tBaseCam1[0], -0.};
    mat3_mul(Hn, M, Q);
}

// Triangulator::regDiv  triangulator.cpp:114-128
VG_HD double tri_reg_div(double num, double denom, double eps)
{
    if (denom > eps * num) return num / denom;
    else if (num == 0) return 2. / eps;
    else return 2. / eps - denom / (num * eps * eps);
}

VG_HD double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// Triangulator::computeRegular with res1 / jac1 only (what SparseReprojectCost asks for, :295-300); jac1 may be NULL
VG_HD double triangulate_regular(const double *R, const double *t, double eps, const double *p, const double *q_in, double *jac1)
{
    double q[3], r[3];
    mat3_vec(R, q_in, q);
#pragma unroll
    for (int i = 0; i < 3; i++) r[i] = p[i] + q[i];
    const double tp = dot3(t, p), tq = dot3(t, q), tr = dot3(t, r), tt = dot3(t, t), rp = dot3(r, p), rq = dot3(r, q);
    const double delta = tp * rq - tq * rp;
    const double delta1 = tt * rq - tr * tq;
    const double res1 = tri_reg_div(delta1, delta, eps);
    if (jac1) {
        const double deltaInv = 1. / delta;
        const double qSkew[9] = {0, -q[2], q[1], q[2], 0, -q[0], -q[1], q[0], 0};
        double dV[3], dO[3], d1V[3], d1O[3], w[3];
#pragma unroll
        for (int i = 0; i < 3; i++) dV[i] = rq * p[i] - rp * q[i];
#pragma unroll
        for (int i = 0; i < 3; i++) w[i] = tp * p[i] - tq * p[i] - rp * t[i];
        mat3_vec(qSkew, w, dO);
#pragma unroll
        for (int i = 0; i < 3; i++) d1V[i] = 2 * rq * t[i] - tq * r[i] - tr * q[i];
#pragma unroll
        for (int i = 0; i < 3; i++) w[i] = tt * p[i] - tq * t[i] - tr * t[i];
        mat3_vec(qSkew, w, d1O);
        if (delta > eps * delta1) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                jac1[i] = deltaInv * (d1V[i] - res1 * dV[i]);
                jac1[3 + i] = deltaInv * (d1O[i] - res1 * dO[i]);
            }
        } else if (delta1 == 0) {
#pragma unroll
            for (int i = 0; i < 6; i++) jac1[i] = 0.;
        } else {
            const double coef = 1. / (eps * eps);
            const double deltaInv1 = 1. / delta1;
            const double k = -coef * deltaInv1;
            const double k1 = coef * delta * deltaInv1 * deltaInv1;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                jac1[i] = k * dV[i] + k1 * d1V[i];
                jac1[3 + i] = k * dO[i] + k1 * d1O[i];
            }
        }
    }
    return res1;
}

// ------------------------------------------------------------------------------------------ kernels
// one lane per block: the block's frame at its current odometry parameter
#ifdef VG_TU_LOCAL  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(64) void vg_local_frame_kernel(const double *__restrict__ xiBaseCam, const double *__restrict__ base,
                                                             const double *__restrict__ xiOdom, long long first_block, long long n_blocks,
                                                             int sparse, double *__restrict__ frames)
{
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const double *xo = xiOdom + 6 * b;                 // parameter array of THIS launch: block first_block + b at row b
    if (sparse) sparse_frame_c(xiBaseCam, base, xo, frames + (first_block + b) * kSparseFrame);
    else mono_frame_c(xiBaseCam, base, xo, frames + (first_block + b) * kMonoFrame);
}

// once per set, at its creation: the base constants (one lane)
__global__ __launch_bounds__(64) void vg_local_base_kernel(const double *__restrict__ xiBaseCam, double *__restrict__ base)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) base_const(xiBaseCam, base);
}
#endif

struct MonoArgs {
    const double *xb;       // xiBaseCam [6]
    const double *base;     // [kBaseConst] what the frames need of xiBaseCam alone (vg_local_base_kernel)
    const double *xi_odom;  // rows of this launch: [count][6]
    const double *intr;
    const double *x1;       // [n_blocks][5][3]
    const double *p2;       // [n_blocks][5][2]
    const double *lengths;  // rows of this launch: [count][5]
    double *res;            // [count][10]
    double *jac_odom;       // [count][10][6] or NULL
    double *jac_len;        // [count][10][5] or NULL
    long long first_block;
    unsigned int n_points;  // count * 5
};

// A workgroup owns 64 five-point blocks = 320 points: their frames are computed IN the kernel (first wave, one lane per
// block, into LDS) -- with 5 points per 34-double frame a frame launch of its own writes and re-reads 45 % of the algorithmic bytes
// on top (1 M features: 103 us in two launches, see profiles/NOTES.md).  The inputs of the point phase are requested before
// the frame phase, so their latency hides behind it.
constexpr int kMonoThreads = kEmitThreads;
constexpr int kMonoWgBlocks = (kMonoThreads + kMonoPoints - 1) / kMonoPoints + 2;
constexpr int kMonoLdsDoubles = (kMonoThreads / kWave) * 2 * kWave * 6 + kMonoWgBlocks * kMonoFrame;

template <int MODEL>
__global__ __launch_bounds__(kMonoThreads) void vg_mono_reproject_kernel(MonoArgs a)
{
    constexpr int K = CameraTraits<MODEL>::K;
    using d2 = HIP_vector_type<double, 2>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *lds_frames = smem + (kMonoThreads / kWave) * 2 * kWave * 6;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const unsigned int o0 = blockIdx.x * (unsigned)kMonoThreads, o = o0 + tid;
    const bool active = o < a.n_points;
    const unsigned int oc = active ? o : a.n_points - 1;
    const unsigned int bl = oc / kMonoPoints, i = oc - bl * kMonoPoints;   // block of this launch, point of the block
    const long long b = a.first_block + bl;
    // frames of the workgroup's blocks
    const unsigned int bl0 = o0 / kMonoPoints;
    {
        const unsigned int last = (o0 + kMonoThreads - 1 < a.n_points ? o0 + kMonoThreads - 1 : a.n_points - 1) / kMonoPoints;
        if ((unsigned)tid <= last - bl0) {
            double fr[kMonoFrame];
            mono_frame_c(a.xb, a.base, a.xi_odom + 6 * (size_t)(bl0 + tid), fr);
            fr[kMonoFrame - 1] = 0.;
#pragma unroll
            for (int k = 0; k < kMonoFrame; k++) lds_frames[tid * kMonoFrame + k] = fr[k];
        }
    }
    __syncthreads();
    const double *f = lds_frames + (bl - bl0) * kMonoFrame;
    const double *x1p = a.x1 + (b * kMonoPoints + i) * 3;
    const double x1[3] = {x1p[0], x1p[1], x1p[2]};
    const double len = a.lengths[(size_t)bl * kMonoPoints + i];
    const d2 ob = reinterpret_cast<const d2 *>(a.p2)[b * kMonoPoints + i];
    // xVec2[i] = _xVec1[i] * params[1][i]; xi21.transform: R * x + t   (:224-229)
    const double s[3] = {x1[0] * len, x1[1] * len, x1[2] * len};
    double rx[3];
    mat3_vec(f + 3, s, rx);
    const double X0 = rx[0] + f[0], X1 = rx[1] + f[1], X2 = rx[2] + f[2];
    CornerEval<K> e;
    eval_corner<MODEL, true, false>(a.intr, X0, X1, X2, e);
    d2 r;
    r.x = e.ok ? e.u - ob.x : kDoubleBig;   // :232-244
    r.y = e.ok ? e.v - ob.y : kDoubleBig;
    if (active) reinterpret_cast<d2 *>(a.res)[o] = r;
    const unsigned int ow = o0 + wave * kWave;
    int n_valid = 0;
    if (ow < a.n_points) n_valid = (a.n_points - ow < (unsigned)kWave) ? (int)(a.n_points - ow) : kWave;
    double *stage = smem + wave * (2 * kWave * 6);
    if (a.jac_len) {
        // rows 2i / 2i + 1 of the [10 x 5] block: zero but for column i  (:262-275); streamed out through the wave's tile
        double n2[3];
        mat3_vec(f + 3, x1, n2);
        const double du = e.P[0] * n2[0] + e.P[1] * n2[1] + e.P[2] * n2[2];
        const double dv = e.P[3] * n2[0] + e.P[4] * n2[1] + e.P[5] * n2[2];
        double row[2 * kMonoPoints];
#pragma unroll
        for (int c = 0; c < kMonoPoints; c++) {
            row[c] = c == (int)i ? du : 0.;
            row[kMonoPoints + c] = c == (int)i ? dv : 0.;
        }
        wave_store_rows<kMonoPoints>(stage, row, a.jac_len + (size_t)ow * 2 * kMonoPoints, n_valid, lane);
    }
    if (a.jac_odom) {
        double rows[12];
        pose_rows(e.P, X0, X1, X2, f + 12, rows);   // InterJacobian::dpdxi, failed EUCM points: P = 0 -> zero rows
        wave_store_rows<6>(stage, rows, a.jac_odom + (size_t)ow * 12, n_valid, lane);
    }
}

struct SparseArgs {
    const double *frames;      // [n_blocks][kSparseFrame]
    const double *intr;
    const double *x1, *x2;     // [total][3]
    const double *p2;          // [total][2]
    const double *size;        // [total]
    const int *point_block;    // [total] block of every point
    const double *base;        // [kBaseConst] what the frames need of xiBaseCam alone (vg_local_base_kernel)
    const double *xb;          // xiBaseCam [6]                      } the FUSED kernel computes the frames itself
    const double *xi_odom;     // rows of this launch: [n_blocks][6] }
    long long first_block;     // block of row 0 of xi_odom
    double *res;               // rows of this launch: [count][2]
    double *jac;               // [count][2][6] or NULL
    long long first_point;
    unsigned int n_points;     // count
};

// one point of a SparseReprojectCost block; f = the block's frame (a wave-uniform or a per-lane address)
template <int MODEL>
__device__ __forceinline__ void sparse_point(const SparseArgs &a, const double *__restrict__ f, const long long pt, const unsigned int o,
                                             const bool active, const unsigned int o0, double *stage, const int wave, const int lane)
{
    constexpr int K = CameraTraits<MODEL>::K;
    using d2 = HIP_vector_type<double, 2>;
    const double *x1 = a.x1 + pt * 3, *x2 = a.x2 + pt * 3;
    const double *t12 = f, *Rt = f + 3, *R21 = f + 12;
    const bool want_jac = a.jac != nullptr;
    double jv[6];
    // Triangulator(xi12): eps = 1e-3.  (Two calls: a pointer chosen at run time would put jv into scratch memory.)
    const double lam = want_jac ? triangulate_regular(Rt, t12, 1e-3, x1, x2, jv) : triangulate_regular(Rt, t12, 1e-3, x1, x2, nullptr);
    // xVec2 = xVec1 * lambda; xi12.inverseTransform: R(-r) * (x - t)   (:303-308)
    const double d[3] = {x1[0] * lam - t12[0], x1[1] * lam - t12[1], x1[2] * lam - t12[2]};
    double X[3];
    mat3_vec(R21, d, X);
    CornerEval<K> e;
    eval_corner<MODEL, true, false>(a.intr, X[0], X[1], X[2], e);
    const d2 ob = reinterpret_cast<const d2 *>(a.p2)[pt];
    // the reference divides two residuals and six Jacobian entries by the feature size; here ONE division and eight products
    // (each within an ulp of the quotient; an IEEE FP64 division is eleven instructions on this part)
    const double rsz = 1. / a.size[pt];
    d2 r;
    r.x = e.ok ? (e.u - ob.x) * rsz : kDoubleBig;   // :311-323
    r.y = e.ok ? (e.v - ob.y) * rsz : kDoubleBig;
    if (active) reinterpret_cast<d2 *>(a.res)[o] = r;
    if (want_jac) {
        double rows[12];
        pose_rows(e.P, X[0], X[1], X[2], f + 21, rows);   // :331-345 (a failed point has P = 0: zero rows, as :336-339)
        // depth Jacobian (:360-381)
        const double *RcamBase = f + 42, *M = f + 51, *Q = f + 60;
        double n2[3];
        mat3_vec(R21, x1, n2);
        const double dpdl0 = e.P[0] * n2[0] + e.P[1] * n2[1] + e.P[2] * n2[2];
        const double dpdl1 = e.P[3] * n2[0] + e.P[4] * n2[1] + e.P[5] * n2[2];
        const double *dldv = jv, *dldw = jv + 3;
        double dldt[3], dldr[3];
#pragma unroll
        for (int j = 0; j < 3; j++) dldt[j] = dldv[0] * RcamBase[0 + j] + dldv[1] * RcamBase[3 + j] + dldv[2] * RcamBase[6 + j];
#pragma unroll
        for (int j = 0; j < 3; j++)
            dldr[j] = (dldw[0] * M[0 + j] + dldw[1] * M[3 + j] + dldw[2] * M[6 + j]) + (dldv[0] * Q[0 + j] + dldv[1] * Q[3 + j] + dldv[2] * Q[6 + j]);
        if (e.ok) {   // `if (residual[2*i] == DOUBLE_BIG) continue;`  :362
#pragma unroll
            for (int j = 0; j < 3; j++) {
                rows[j] += dpdl0 * dldt[j];
                rows[3 + j] += dpdl0 * dldr[j];
                rows[6 + j] += dpdl1 * dldt[j];
                rows[9 + j] += dpdl1 * dldr[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 12; j++) rows[j] = 0.;   // :336-339 (UCM / Mei never fail; EUCM rows are zero already)
        }
        // :383-389: ONLY the u-row is divided by the feature size
#pragma unroll
        for (int j = 0; j < 6; j++) rows[j] *= rsz;
        const unsigned int ow = o0 + wave * kWave;
        int n_valid = 0;
        if (ow < a.n_points) n_valid = (a.n_points - ow < (unsigned)kWave) ? (int)(a.n_points - ow) : kWave;
        wave_store_rows<6>(stage, rows, a.jac + (size_t)ow * 12, n_valid, lane);
    }
}

// FUSED: the frames of the blocks this workgroup's points belong to are computed in the kernel (one lane per block, into
// LDS behind the store tiles) -- for the launches the reference actually makes per frame pair (200 RANSAC hypotheses of a few
// points, then one block of all inliers: a handful of workgroups) the frame launch is half of the evaluation's latency.
// The host takes this route when the grid is small and no workgroup touches more than kSparseWgBlocks blocks.
constexpr int kSparseWgBlocks = 40;

#ifndef VG_SPARSE_WAVES
#define VG_SPARSE_WAVES 4   // waves per SIMD the register allocation aims at (experiment switch; see profiles/NOTES.md)
#endif
template <int MODEL, bool FUSED>
__global__ __launch_bounds__(kEmitThreads) __attribute__((amdgpu_waves_per_eu(VG_SPARSE_WAVES, 8))) void vg_sparse_reproject_kernel(SparseArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const unsigned int o0 = blockIdx.x * (unsigned)kEmitThreads, o = o0 + tid;
    const bool active = o < a.n_points;
    const long long pt = a.first_point + (active ? o : a.n_points - 1);
    double *stage = smem + wave * (2 * kWave * 6);
    const int blk = a.point_block[pt];
    if constexpr (FUSED) {
        double *lds_frames = smem + (kEmitThreads / kWave) * 2 * kWave * 6;
        const unsigned int o_last = o0 + kEmitThreads - 1 < a.n_points ? o0 + kEmitThreads - 1 : a.n_points - 1;
        const int b_first = a.point_block[a.first_point + o0], b_last = a.point_block[a.first_point + o_last];
        if (tid <= b_last - b_first)   // blocks without points in between cost a lane each and are never read
            sparse_frame_c(a.xb, a.base, a.xi_odom + 6 * (size_t)(b_first + tid - a.first_block), lds_frames + tid * kSparseFrame);
        __syncthreads();
        sparse_point<MODEL>(a, lds_frames + (blk - b_first) * kSparseFrame, pt, o, active, o0, stage, wave, lane);
    } else {
        // The 69 doubles of the block's frame: when all 64 points of the wave belong to ONE block (blocks of hundreds of
        // inliers) the frame address is wave-uniform and its loads are scalar -- one fetch for the wave instead of 69 vector
        // loads of the same address in every lane; a wave that straddles blocks reads per lane.  Same arithmetic on both paths.
        const int blk0 = __builtin_amdgcn_readfirstlane(blk);
        if (__builtin_amdgcn_ballot_w64(blk != blk0) == 0) sparse_point<MODEL>(a, a.frames + (long long)blk0 * kSparseFrame, pt, o, active, o0, stage, wave, lane);
        else sparse_point<MODEL>(a, a.frames + (long long)blk * kSparseFrame, pt, o, active, o0, stage, wave, lane);
    }
}

// CameraJacobian: L11 | L12 | L22 (jacobian.h:54-71), computed on the host once per call
struct CameraJacobianArgs {
    double L11[9], L12[9], L22[9];
    int two;                // twoTransforms
    double intr[10];        // by value: the camera of this call
    const double *X2;       // [n][3]
    const double *grad;     // [n][2] or NULL
    double *dpdxi;          // [n][2][6] (u-row, v-row) or NULL
    double *dfdxi;          // [n][6] or NULL
    unsigned int n;
};

VG_HD void camera_jacobian_frame(const double *T12, const double *T23, double *L11, double *L12, double *L22)
{
    double M[9];
    const RotTrig g = rot_trig(T12 + 3, true, true);
    inter_omega_rot(T12 + 3, g, M);
    if (T23) {
        double R21[9], R32[9], nR32[9], a[9], b[9];
        rotation_matrix(T12 + 3, -1., g, R21);
        transf_rot_mat(T23, -1., R32);
        mat3_mul(R32, R21, L11);
        mat3_mul(L11, M, L22);
#pragma unroll
        for (int i = 0; i < 9; i++) nR32[i] = -R32[i];
        const double H[9] = {0, -T23[2], T23[1], T23[2], 0, -T23[0], -T23[1], T23[0], 0};
        mat3_mul(nR32, H, a);
        mat3_mul(a, R21, b);
        mat3_mul(b, M, L12);   // L12 = -R32 * hat(T23.trans()) * R21 * M
    } else {
        rotation_matrix(T12 + 3, -1., g, L11);
#pragma unroll
        for (int i = 0; i < 9; i++) L12[i] = 0.;
        mat3_mul(L11, M, L22);
    }
}

template <int MODEL>
__global__ __launch_bounds__(kEmitThreads) void vg_camera_jacobian_kernel(CameraJacobianArgs a)
{
    constexpr int K = CameraTraits<MODEL>::K;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const unsigned int o0 = blockIdx.x * (unsigned)kEmitThreads, ip = o0 + tid;
    const bool active = ip < a.n;
    const size_t i = active ? ip : a.n - 1;   // the whole wave goes through the store tile; surplus lanes recompute the last point
    const double X[3] = {a.X2[3 * i], a.X2[3 * i + 1], a.X2[3 * i + 2]};
    CornerEval<K> e;
    eval_corner<MODEL, true, false>(a.intr, X[0], X[1], X[2], e);
    // B = hat(X2) * L22 (- L12)   jacobian.h:87
    const double H[9] = {0, -X[2], X[1], X[2], 0, -X[0], -X[1], X[0], 0};
    double B[9];
    mat3_mul(H, a.L22, B);
    if (a.two) {
#pragma unroll
        for (int k = 0; k < 9; k++) B[k] = B[k] - a.L12[k];
    }
    const unsigned int ow = o0 + wave * kWave;
    int n_valid = 0;
    if (ow < a.n) n_valid = (a.n - ow < (unsigned)kWave) ? (int)(a.n - ow) : kWave;
    double *stage = smem + wave * (2 * kWave * 6);
    // UCM / Mei never report failure; a failed EUCM point has P = 0 and the rows come out zero (with the sign of zero
    // the reference's fill(0.) does not have: written as +0 below).  Rows go out through the wave's store tile (16-byte
    // pieces at a 96 / 48-byte lane stride otherwise: 0.38 of the HBM peak for 1 M points).
    if (a.dpdxi) {
        double out[12];
#pragma unroll
        for (int row = 0; row < 2; row++) {
            const double *p = e.P + 3 * row;
            const double n[3] = {-p[0], -p[1], -p[2]};
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double tr = n[0] * a.L11[0 + j] + n[1] * a.L11[3 + j] + n[2] * a.L11[6 + j];   // -row * L11
                const double ro = p[0] * B[0 + j] + p[1] * B[3 + j] + p[2] * B[6 + j];               // row * B
                out[6 * row + j] = e.ok ? tr : 0.;
                out[6 * row + 3 + j] = e.ok ? ro : 0.;
            }
        }
        wave_store_rows<6>(stage, out, a.dpdxi + (size_t)ow * 12, n_valid, lane);
    }
    if (a.dfdxi && a.grad) {
        const double g0 = a.grad[2 * i], g1 = a.grad[2 * i + 1];
        double d[3], n[3], out[6];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            d[j] = g0 * e.P[j] + g1 * e.P[3 + j];   // dfdX = grad * projJac
            n[j] = -d[j];
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double tr = n[0] * a.L11[0 + j] + n[1] * a.L11[3 + j] + n[2] * a.L11[6 + j];
            const double ro = d[0] * B[0 + j] + d[1] * B[3 + j] + d[2] * B[6 + j];
            out[j] = e.ok ? tr : 0.;
            out[3 + j] = e.ok ? ro : 0.;
        }
        wave_store_rows<3>(stage, out, a.dfdxi + (size_t)ow * 6, n_valid, lane);
    }
}

}  // namespace vg
