// vg_transf_host.hpp -- host-side pieces of Transformation<double> (include/geometry/transformation.h) used by the
// front end (pose initialisation, residual report) and by the solver's prior blocks.  Same arithmetic as the device
// code: they are the vg_geometry.hpp functions called on the host.
#pragma once

#include <array>
#include <cmath>

#include "vg_geometry.hpp"

namespace vgth {

using Array6d = std::array<double, 6>;  // include/std.h:43

inline vg::Quat quat_of(const double *rot)
{
    const vg::RotTrig g = vg::rot_trig(rot, false, true);
    return vg::quat_from_rotvec(rot, g);
}

// Transformation::compose  transformation.h:80-88
inline Array6d compose(const Array6d &a, const Array6d &b)
{
    const vg::Quat q1 = quat_of(a.data() + 3), q2 = quat_of(b.data() + 3);
    double rt[3];
    vg::quat_rotate(q1, b.data(), rt);
    Array6d r;
    for (int i = 0; i < 3; i++) r[i] = rt[i] + a[i];
    vg::quat_to_rotvec(vg::quat_mul(q1, q2), r.data() + 3);
    return r;
}

// Transformation::inverseCompose  transformation.h:90-99   (a^-1 o b)
inline Array6d inverse_compose(const Array6d &a, const Array6d &b)
{
    const vg::Quat q1 = quat_of(a.data() + 3), q2 = quat_of(b.data() + 3);
    const vg::Quat q1inv = {-q1.x, -q1.y, -q1.z, q1.w};
    const double d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    Array6d r;
    vg::quat_rotate(q1inv, d, r.data());
    vg::quat_to_rotvec(vg::quat_mul(q1inv, q2), r.data() + 3);
    return r;
}

// Transformation::composeInverse  transformation.h:101-110   (a o b^-1)
inline Array6d compose_inverse(const Array6d &a, const Array6d &b)
{
    const vg::Quat q1 = quat_of(a.data() + 3), q2 = quat_of(b.data() + 3);
    const vg::Quat q2inv = {-q2.x, -q2.y, -q2.z, q2.w};
    const vg::Quat qres = vg::quat_mul(q1, q2inv);
    double rt[3];
    vg::quat_rotate(qres, b.data(), rt);
    Array6d r;
    for (int i = 0; i < 3; i++) r[i] = a[i] - rt[i];
    vg::quat_to_rotvec(qres, r.data() + 3);
    return r;
}

// Transformation::inverse  transformation.h:112-119
inline Array6d inverse(const Array6d &a)
{
    const double neg[3] = {-a[3], -a[4], -a[5]};
    const vg::RotTrig g = vg::rot_trig(a.data() + 3, true, false);
    double R[9];
    vg::rotation_matrix(a.data() + 3, -1., g, R);  // rotMatInv
    Array6d r;
    for (int i = 0; i < 3; i++) r[i] = -(R[3 * i] * a[0] + R[3 * i + 1] * a[1] + R[3 * i + 2] * a[2]);
    r[3] = neg[0]; r[4] = neg[1]; r[5] = neg[2];
    return r;
}

// rotationVector(R) = Quaternion(R).toRotationVector()   geometry_core.h:120-124, quaternion.h:52-59
// (assumes 1 + trace(R) > 0, like the reference: SURVEY D10)
inline void rotvec_from_matrix(const double *R, double *rot)
{
    vg::Quat q;
    q.w = std::sqrt(1.0 + (R[0] + R[4] + R[8])) / 2.0;
    const double w4 = 4.0 * q.w;
    q.x = (R[7] - R[5]) / w4;
    q.y = (R[2] - R[6]) / w4;
    q.z = (R[3] - R[1]) / w4;
    vg::quat_to_rotvec(q, rot);
}

}  // namespace vgth
