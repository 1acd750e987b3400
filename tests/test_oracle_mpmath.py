"""The oracle (oracle/vg_oracle.c, double precision) against a SECOND, independent restatement of the reference's
formulas -- written from the reference sources with their first-order branches and thresholds, evaluated with mpmath at
50 digits.  Two hands, two languages, two precisions: a transcription slip in one of them shows up here at the 1e-12
level (the float64 autograd check of tests/test_oracle_math.py reformulates through rotation
matrices and could only be held to 1e-9).  Covers GenericProjectionJac::Evaluate for the three models and chains of
length 0..5, TransformationPrior and OdometryPrior.

Every function cites the reference lines it restates (paths relative to /root/reference).  Test points stay 1e-3
(relative) away from the branch thresholds: there the double and the 50-digit evaluation could legitimately take
different branches; the thresholds themselves are exercised by tests/test_oracle_math.py and the GPU fuzz slice.
"""
import numpy as np
import pytest

mp = pytest.importorskip("mpmath")
from mpmath import mpf  # noqa: E402

from oracle import vgo  # noqa: E402

mp.mp.dps = 50
TOL = 1e-12
M_PI = mpf(float(np.pi))  # the reference's T_PI is the double constant


def V(a):
    return [mpf(float(x)) for x in np.asarray(a, float).ravel()]


def norm(v):
    return mp.sqrt(sum(x * x for x in v))


def matmul(A, B):
    return [[sum(A[i][k] * B[k][j] for k in range(len(B))) for j in range(len(B[0]))] for i in range(len(A))]


def matvec(A, v):
    return [sum(A[i][k] * v[k] for k in range(len(v))) for i in range(len(A))]


def transpose(A):
    return [list(r) for r in zip(*A)]


def hat(u):  # include/geometry/geometry_core.h:126-132
    return [[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]]


def sinc(x):  # geometry_core.h:24-30
    return mpf(1) if x == 0 else mp.sin(x) / x


def rotation_matrix(v):  # geometry_core.h:40-76
    th = norm(v)
    if th < mpf("1e-5"):
        return [[1, -v[2], v[1]], [v[2], 1, -v[0]], [-v[1], v[0], 1]]
    u = [x / th for x in v]
    s, c = mp.sin(th), 1 - mp.cos(th)
    R = [[0] * 3 for _ in range(3)]
    for i in range(3):
        R[i][i] = 1 + c * (u[i] * u[i] - 1)
    R[0][1] = -s * u[2] + c * u[0] * u[1]
    R[0][2] = s * u[1] + c * u[0] * u[2]
    R[1][2] = -s * u[0] + c * u[1] * u[2]
    R[1][0] = s * u[2] + c * u[1] * u[0]
    R[2][0] = -s * u[1] + c * u[2] * u[0]
    R[2][1] = s * u[0] + c * u[2] * u[1]
    return R


def inter_omega_rot(v):  # geometry_core.h:158-180
    th = norm(v)
    if th < mpf("1e-5"):
        h = [x / 2 for x in v]
        return [[1, -h[2], h[1]], [h[2], 1, -h[0]], [-h[1], h[0], 1]]
    uh = hat([x / th for x in v])
    half = th / 2
    K1 = half * sinc(half) ** 2
    K2 = 1 - sinc(th)
    uu = matmul(uh, uh)
    return [[(1 if i == j else 0) + K1 * uh[i][j] + K2 * uu[i][j] for j in range(3)] for i in range(3)]


class Quat:  # include/geometry/quaternion.h
    def __init__(self, x, y, z, w):
        self.x, self.y, self.z, self.w = x, y, z, w

    @staticmethod
    def from_rotvec(r):  # :31-50
        th = norm(r)
        if abs(th) < mpf("1e-6"):
            return Quat(r[0] / 2, r[1] / 2, r[2] / 2, mpf(1))
        s = mp.sin(th / 2)
        return Quat(r[0] / th * s, r[1] / th * s, r[2] / th * s, mp.cos(th / 2))

    def rotate(self, v):  # :61-82
        x, y, z, w = self.x, self.y, self.z, self.w
        t1, t2, t3, t4, t5, t6, t7, t8, t9 = w * x, w * y, w * z, -x * x, x * y, x * z, -y * y, y * z, -z * z
        return [2 * ((t7 + t9) * v[0] + (t5 - t3) * v[1] + (t2 + t6) * v[2]) + v[0],
                2 * ((t3 + t5) * v[0] + (t4 + t9) * v[1] + (t8 - t1) * v[2]) + v[1],
                2 * ((t6 - t2) * v[0] + (t1 + t8) * v[1] + (t4 + t7) * v[2]) + v[2]]

    def to_rotvec(self):  # :84-98, normalizeAngle geometry_core.h:32-38
        s = mp.sqrt(self.x ** 2 + self.y ** 2 + self.z ** 2)
        u = [self.x, self.y, self.z]
        if s < mpf("1e-5"):
            return [2 * a for a in u]
        th = 2 * mp.atan2(s, self.w)
        if th > M_PI:
            th -= 2 * M_PI
        elif th < -M_PI:
            th += 2 * M_PI
        return [a / s * th for a in u]

    def inv(self):  # :100-103
        return Quat(-self.x, -self.y, -self.z, self.w)

    def __mul__(self, q):  # :105-118
        x, y, z, w = self.x, self.y, self.z, self.w
        return Quat(w * q.x + x * q.w + y * q.z - z * q.y, w * q.y - x * q.z + y * q.w + z * q.x,
                    w * q.z + x * q.y - y * q.x + z * q.w, w * q.w - x * q.x - y * q.y - z * q.z)


class Transf:  # include/geometry/transformation.h, data = [t(3), r(3)] (:46)
    def __init__(self, t=None, r=None):
        self.t = list(t) if t is not None else [mpf(0)] * 3
        self.r = list(r) if r is not None else [mpf(0)] * 3

    @staticmethod
    def from_data(d):
        d = V(d)
        return Transf(d[:3], d[3:])

    def compose(self, o):  # :80-88
        q1, q2 = Quat.from_rotvec(self.r), Quat.from_rotvec(o.r)
        t = [a + b for a, b in zip(q1.rotate(o.t), self.t)]
        return Transf(t, (q1 * q2).to_rotvec())

    def inverse_compose(self, o):  # :90-99
        q1, q2 = Quat.from_rotvec(self.r), Quat.from_rotvec(o.r)
        qi = q1.inv()
        return Transf(qi.rotate([a - b for a, b in zip(o.t, self.t)]), (qi * q2).to_rotvec())

    def compose_inverse(self, o):  # :101-110
        q1, q2 = Quat.from_rotvec(self.r), Quat.from_rotvec(o.r)
        qres = q1 * q2.inv()
        return Transf([a - b for a, b in zip(self.t, qres.rotate(o.t))], qres.to_rotvec())

    def rot_mat(self):  # :131
        return rotation_matrix(self.r)

    def rot_mat_inv(self):  # :132
        return rotation_matrix([-a for a in self.r])

    def screw_transf_inv(self):  # :234-243
        R = self.rot_mat_inv()
        RH = matmul(R, hat(self.t))
        S = [[mpf(0)] * 6 for _ in range(6)]
        for i in range(3):
            for j in range(3):
                S[i][j] = R[i][j]
                S[i][3 + j] = -RH[i][j]
                S[3 + i][3 + j] = R[i][j]
        return S

    def inverse(self):  # :112-119
        R = self.rot_mat_inv()
        return Transf([-sum(R[i][k] * self.t[k] for k in range(3)) for i in range(3)], [-a for a in self.r])

    def array(self):
        return self.t + self.r


# ------------------------------------------------------------------------------------------ camera models
def eucm(p, X):  # include/projection/eucm.h:29-63, 115-167, 169-226
    al, be, fu, fv, u0, v0 = p
    x, y, z = X
    rho = mp.sqrt(z * z + be * (x * x + y * y))
    eta = al * rho + (1 - al) * z
    ok = not (eta < mpf("1e-3"))
    if ok and al > mpf("0.5") and z / eta < (al - 1) / (al + al - 1):
        ok = False
    if not ok:
        return False, None, [[mpf(0)] * 3] * 2, [[mpf(0)] * 6] * 2
    uv = [fu * (x / eta) + u0, fv * (y / eta) + v0]
    ga = 1 - al
    k = 1 / eta / eta
    ab = al * be / rho
    Jxy = k * ab * x * y
    Jz = k * (ga + al * z / rho)
    Jx = ga * z + al * rho
    P = [[fu * k * (Jx - ab * x * x), -fu * Jxy, -fu * x * Jz], [-fv * Jxy, fv * k * (Jx - ab * y * y), -fv * y * Jz]]
    s = x * x + y * y
    e2 = eta * eta
    Ji = [[-fu * x * (rho - z) / e2, -fu * x * al * s / (2 * e2 * rho), x / eta, 0, 1, 0],
          [-fv * y * (rho - z) / e2, -fv * y * al * s / (2 * e2 * rho), 0, y / eta, 0, 1]]
    return True, uv, P, Ji


def unified_dm(xi, X):  # ucm.h:120-142 == mei.h:136-156
    x, y, z = X
    rho = mp.sqrt(x * x + y * y + z * z)
    ri = 1 / rho
    di = 1 / (xi * rho + z)
    d2 = di * di
    dm = [[(xi * rho + z - xi * x * x * ri) * d2, -xi * x * y * ri * d2, -x * (1 + xi * z * ri) * d2],
          [-xi * x * y * ri * d2, (xi * rho + z - xi * y * y * ri) * d2, -y * (1 + xi * z * ri) * d2]]
    return rho, di, x * di, y * di, dm


def ucm(p, X):  # include/projection/ucm.h:32-59, 112-151, 153-197
    xi, fu, fv, u0, v0 = p
    rho, di, xn, yn, dm = unified_dm(xi, X)
    uv = [fu * xn + u0, fv * yn + v0]
    P = [[fu * a for a in dm[0]], [fv * a for a in dm[1]]]
    Ji = [[-fu * xn * di * rho, xn, 0, 1, 0], [-fv * yn * di * rho, 0, yn, 0, 1]]
    return True, uv, P, Ji


def mei(p, X):  # include/projection/mei.h:29-68, 121-191, 193-285
    xi, k1, k2, k3, k4, k5, fu, fv, u0, v0 = p
    rho, di, xn, yn, dm = unified_dm(xi, X)
    xx, xy, yy = xn * xn, xn * yn, yn * yn
    r2 = xx + yy
    D = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    dx = 2 * k4 * xy + k5 * (r2 + 2 * xx)
    dy = 2 * k5 * xy + k4 * (r2 + 2 * yy)
    xd, yd = xn * D + dx, yn * D + dy
    uv = [fu * xd + u0, fv * yd + v0]
    dD = k1 + 2 * k2 * r2 + 3 * k3 * r2 ** 2
    a = [fu * (D + 2 * xx * dD + 2 * k4 * yn + 6 * k5 * xn), fu * (2 * xy * dD + 2 * k4 * xn + 2 * k5 * yn)]
    b = [fv * (2 * xy * dD + 2 * k5 * yn + 2 * k4 * xn), fv * (D + 2 * yy * dD + 2 * k5 * xn + 6 * k4 * yn)]
    P = [[a[0] * dm[0][j] + a[1] * dm[1][j] for j in range(3)], [b[0] * dm[0][j] + b[1] * dm[1][j] for j in range(3)]]
    dxi, dyi = -xn * di * rho, -yn * di * rho
    Ji = [[a[0] * dxi + a[1] * dyi, fu * xn * r2, fu * xn * r2 ** 2, fu * xn * r2 ** 3, 2 * fu * xy, fu * (r2 + 2 * xx), xd, 0, 1, 0],
          [b[0] * dxi + b[1] * dyi, fv * yn * r2, fv * yn * r2 ** 2, fv * yn * r2 ** 3, fv * (r2 + 2 * yy), 2 * fv * xy, 0, yd, 0, 1]]
    return True, uv, P, Ji


CAMERAS = {"eucm": eucm, "ucm": ucm, "mei": mei}
BIG = mpf("1e15")  # include/std.h:71


def evaluate(model, status, grid, obs, params):
    """GenericProjectionJac::Evaluate, src/calibration/calib_cost_functions.cpp:28-117, with every Jacobian block"""
    return evaluate_mp(model, status, grid, obs, [V(b) for b in params])


def evaluate_mp(model, status, grid, obs, params):
    """the same with the parameter blocks already in 50 digits (lists of mpf)"""
    cam = CAMERAS[model]
    intr = list(params[0])
    members = [Transf(list(q[:3]), list(q[3:])) for q in params[1:]]
    acc = Transf()
    frames = []  # per member: (R12, M12, t13)   InterJacobian ctor, include/projection/jacobian.h:139-152
    for xi23, st in zip(members, status):
        if st == 0:
            acc = acc.compose(xi23)
            xi13 = acc
        else:
            xi13 = acc
            acc = acc.compose_inverse(xi23)
        R12 = matmul(xi13.rot_mat(), xi23.rot_mat_inv())
        M12 = matmul(R12, inter_omega_rot(xi23.r))
        if st == 1:
            R12 = [[-a for a in r] for r in R12]
            M12 = [[-a for a in r] for r in M12]
        frames.append((R12, M12, list(xi13.t)))
    R = acc.rot_mat()
    N = len(grid)
    K = len(intr)
    res = []
    Ji = [[mpf(0)] * K for _ in range(2 * N)]
    Jm = [[[mpf(0)] * 6 for _ in range(2 * N)] for _ in members]
    for i in range(N):
        X = [a + b for a, b in zip(matvec(R, V(grid[i])), acc.t)]  # transformation.h:147-155
        ok, uv, P, JI = cam(intr, X)
        res += [uv[0] - mpf(float(obs[i][0])), uv[1] - mpf(float(obs[i][1]))] if ok else [BIG, BIG]
        for l, (R12, M12, t13) in enumerate(frames):  # dpdxi, jacobian.h:155-171
            H = hat([a - b for a, b in zip(X, t13)])
            for row in range(2):
                tr = matmul([P[row]], R12)[0]
                rot = matmul(matmul([[-a for a in P[row]]], H), M12)[0]
                Jm[l][2 * i + row] = tr + rot
        Ji[2 * i], Ji[2 * i + 1] = list(JI[0]), list(JI[1])
    return res, Ji, Jm


def to_np(a):
    return np.array([[float(x) for x in r] for r in a]) if isinstance(a[0], list) else np.array([float(x) for x in a])


def close(got, ref_mp, what, floor_rel=1e-3):
    """SURVEY 8(c) metric at 1e-12: normwise, and element-wise against max(|ref|, floor_rel * |ref|_inf)"""
    got = np.asarray(got, float)
    ref = to_np(ref_mp)
    # the comparison itself in 50 digits: float(ref) would add its own 1e-16
    flat = [x for r in ref_mp for x in r] if isinstance(ref_mp[0], list) else list(ref_mp)
    diff = np.array([float(mpf(float(g)) - r) for g, r in zip(got.ravel(), flat)]).reshape(got.shape)
    nrm = np.linalg.norm(ref)
    if nrm == 0:
        assert np.all(got == 0), what
        return 0.0
    e1 = np.linalg.norm(diff) / nrm
    e2 = np.max(np.abs(diff) / np.maximum(np.abs(ref), floor_rel * np.max(np.abs(ref))))
    assert e1 <= TOL and e2 <= TOL, "%s: normwise %.2e elementwise %.2e" % (what, e1, e2)
    return max(e1, e2)


RNG = np.random.default_rng(20260929)
INTR = {"eucm": [0.571, 1.18, 312.0, 305.0, 655.0, 391.0], "ucm": [1.31, 702.0, 694.0, 633.0, 409.0],
        "mei": [1.27, -0.04, 0.012, -0.003, 0.0012, -0.0017, 698.0, 705.0, 648.0, 395.0]}


def random_case(model, L, rot_scale=0.6):
    N = 7
    grid = np.column_stack([RNG.uniform(-0.5, 0.5, N), RNG.uniform(-0.35, 0.35, N), RNG.uniform(-0.02, 0.02, N)])
    status = [int(s) for s in RNG.integers(0, 2, L)]
    members = []
    for l in range(L):
        t = RNG.uniform(-0.15, 0.15, 3)
        r = RNG.standard_normal(3)
        r *= rot_scale * RNG.uniform(0.2, 1.0) / np.linalg.norm(r)
        members.append(np.concatenate([t, r]))
    if L:  # keep the board in front of the camera: the last member carries it to z ~ 1
        members[-1][:3] += [0.0, 0.0, 1.0] if status[-1] == 0 else [0.0, 0.0, -1.0]
    else:
        grid[:, 2] += 1.0
    obs = RNG.uniform(100, 1100, (N, 2))
    return grid, obs, status, [np.array(INTR[model])] + members


@pytest.mark.parametrize("model", ["eucm", "ucm", "mei"])
@pytest.mark.parametrize("L", [0, 1, 2, 3, 5])
def test_evaluate_against_50_digit_restatement(model, L):
    worst = 0.0
    for _ in range(3):
        grid, obs, status, params = random_case(model, L)
        r_mp, ji_mp, jm_mp = evaluate(model, status, grid, obs, params)
        r, J = vgo.eval_block(vgo.MODELS[model], status, grid, obs, params)
        # residuals: a difference of near-equal numbers -> measured against the projection, as the parity metric does
        proj = [a + mpf(float(o)) for a, o in zip(r_mp, obs.ravel())]
        dr = np.array([float(mpf(float(g)) - a) for g, a in zip(r, r_mp)])
        assert np.linalg.norm(dr) <= TOL * float(norm(proj)), (model, L, "residual")
        worst = max(worst, close(J[0], ji_mp, "%s L=%d intrinsic block" % (model, L)))
        for l in range(L):
            worst = max(worst, close(J[1 + l], jm_mp[l], "%s L=%d member %d (%s)" % (model, L, l, "ID"[status[l]])))
    print(model, L, "worst normalised difference %.2e" % worst)


@pytest.mark.parametrize("theta", [3e-7, 8e-6, 1.4e-5, 3.1, 3.3, 5.9])
def test_branches_of_the_chain_against_50_digits(theta):
    """each side of the 1e-6 / 1e-5 thresholds and of the +-pi wrap, 1e-3 away from them in relative terms"""
    grid, obs, _, params = random_case("eucm", 2)
    u = np.array([0.3, -0.5, 0.81])
    params[1][3:] = theta * u / np.linalg.norm(u)
    params[2][3:] = 0.7 * theta * np.array([-0.2, 0.9, 0.38]) / np.linalg.norm([-0.2, 0.9, 0.38])
    params[2][:3] = [0.02, -0.03, 1.0 if theta < 1 else -1.0 if 3.0 < theta * 1.7 < 3.3 else 1.0]
    for status in ([0, 0], [1, 0], [0, 1]):
        r_mp, ji_mp, jm_mp = evaluate("eucm", status, grid, obs, params)
        if any(x == BIG for x in r_mp):
            continue  # the composed rotation turned the board away: nothing to compare but the in-band values
        r, J = vgo.eval_block(vgo.MODEL_EUCM, status, grid, obs, params)
        close(J[0], ji_mp, "intrinsic block theta=%g" % theta)
        for l in range(2):
            close(J[1 + l], jm_mp[l], "member %d theta=%g status %s" % (l, theta, status))


def test_failed_projection_is_in_band_in_both():
    grid, obs, _, params = random_case("eucm", 1)
    params[1][:3] = [0, 0, -1.0]
    params[1][3:] = [0.01, 0.02, -0.01]
    r_mp, ji_mp, jm_mp = evaluate("eucm", [0], grid, obs, params)
    r, J = vgo.eval_block(vgo.MODEL_EUCM, [0], grid, obs, params)
    assert all(x == BIG for x in r_mp) and np.all(r == 1e15)
    assert np.all(J[0] == 0) and np.all(J[1] == 0)
    assert all(x == 0 for row in ji_mp for x in row) and all(x == 0 for row in jm_mp[0] for x in row)


def test_transformation_prior_against_50_digits():
    """TransformationPrior ctor include/calibration/calib_cost_functions.h:81-93, Evaluate .cpp:214-228"""
    for _ in range(4):
        st = RNG.uniform(1, 50, 6)
        xp = np.concatenate([RNG.uniform(-0.3, 0.3, 3), RNG.uniform(-0.5, 0.5, 3)])
        x = xp + np.concatenate([RNG.uniform(-0.02, 0.02, 3), RNG.uniform(-0.03, 0.03, 3)])
        prior = Transf.from_data(xp)
        M = inter_omega_rot(prior.r)
        A = [[mpf(0)] * 6 for _ in range(6)]
        for i in range(6):
            A[i][i] = mpf(float(st[i]))
        for i in range(3):  # bottomRightCorner = bottomRightCorner * M
            A[3 + i][3:] = [mpf(float(st[3 + i])) * M[i][j] for j in range(3)]
        R = prior.rot_mat()
        e = prior.inverse_compose(Transf.from_data(x)).array()
        err = matvec(R, e[:3]) + matvec(R, e[3:])
        r, J = vgo.transformation_prior(st, xp, x)
        close(r, matvec(A, err), "prior residual")
        close(J, A, "prior Jacobian")


def cholesky_upper(Ainv):
    """Eigen::LLT<Matrix3d>(A).matrixU(): A = U^T U, U upper triangular with a positive diagonal"""
    n = len(Ainv)
    L = [[mpf(0)] * n for _ in range(n)]
    for i in range(n):
        for j in range(i + 1):
            s = Ainv[i][j] - sum(L[i][k] * L[j][k] for k in range(j))
            L[i][j] = mp.sqrt(s) if i == j else s / L[j][j]
    return transpose(L)


def inv3(A):
    M = mp.matrix(A)
    Mi = M ** -1
    return [[Mi[i, j] for j in range(3)] for i in range(3)]


def test_odometry_prior_against_50_digits():
    """OdometryPrior ctor src/calibration/calib_cost_functions.cpp:119-171, Evaluate :175-212"""
    for _ in range(4):
        o1 = np.concatenate([RNG.uniform(-0.5, 0.5, 2), [0.0, 0.0, 0.0], RNG.uniform(-0.8, 0.8, 1)])
        o2 = o1 + np.concatenate([RNG.uniform(-0.3, 0.3, 2), [0.0, 0.0, 0.0], RNG.uniform(-0.3, 0.3, 1)])
        errV, errW, lam = (float(v) for v in RNG.uniform(0.01, 0.2, 3))
        x1 = o1 + 0.01 * RNG.standard_normal(6)
        x2 = o2 + 0.01 * RNG.standard_normal(6)
        zp = Transf.from_data(o1).inverse_compose(Transf.from_data(o2))
        delta = max(norm(zp.r), mpf("0.01"))
        l = max(norm(zp.t), mpf("0.01"))
        s, c = mp.sin(delta / 2), mp.cos(delta / 2)
        dfdu = [[c, l / 2 * s], [-s, l / 2 * c], [mpf(0), mpf(1)]]
        eV, eW, la = mpf(errV), mpf(errW), mpf(lam)
        Cu = [[max(eV * eV * l * l, mpf("0.01") ** 2), mpf(0)], [mpf(0), max(eW * eW * delta * delta, mpf("0.01") ** 2)]]
        Cx = matmul(matmul(dfdu, Cu), transpose(dfdu))
        for i in range(3):
            Cx[i][i] += la * la
        U = cholesky_upper(inv3(Cx))
        A = [[mpf(0)] * 6 for _ in range(6)]
        for i in range(2):
            A[i][0], A[i][1] = U[i][0], U[i][1]  # _A.topLeftCorner<2, 2>()
            A[i][5] = U[i][2]                    # _A.topRightCorner<2, 1>() = U.topRightCorner<2, 1>(): column 5 of 6
        A[2][2] = 1 / la
        A[3][3] = 1 / la                         # bottomRightCorner<3, 3>() = B
        A[4][4] = 1 / la
        A[5][5] = U[2][2]
        blk = vgo.OdometryPrior(errV, errW, lam, o1, o2)
        close(blk.zeta, zp.array(), "zetaPrior")
        close(blk.A, A, "A", floor_rel=1e-6)
        X1, X2 = Transf.from_data(x1), Transf.from_data(x2)
        zeta = X1.inverse_compose(X2)
        res = matvec(A, zp.inverse_compose(zeta).array())

        def jblock(X):  # [R^-1, 0; 0, R^-1 interOmegaRot(rot)]
            Ri = X.rot_mat_inv()
            RM = matmul(Ri, inter_omega_rot(X.r))
            J = [[mpf(0)] * 6 for _ in range(6)]
            for i in range(3):
                for j in range(3):
                    J[i][j] = Ri[i][j]
                    J[3 + i][3 + j] = RM[i][j]
            return J

        J1 = [[-a for a in row] for row in matmul(matmul(A, zeta.screw_transf_inv()), jblock(X1))]
        J2 = matmul(A, jblock(X2))
        r, j1, j2 = blk.evaluate(x1, x2)
        close(r, res, "odometry residual", floor_rel=1e-6)
        close(j1, J1, "odometry J1")
        close(j2, J2, "odometry J2")


def odometry_A(zp, errV, errW, lam):
    """the weighting matrix both odometry blocks build from their zetaPrior (calib_cost_functions.cpp:127-167,
    odometry_cost_function.cpp:160-194)"""
    delta = max(norm(zp.r), mpf("0.01"))
    l = max(norm(zp.t), mpf("0.01"))
    s, c = mp.sin(delta / 2), mp.cos(delta / 2)
    dfdu = [[c, l / 2 * s], [-s, l / 2 * c], [mpf(0), mpf(1)]]
    eV, eW, la = mpf(errV), mpf(errW), mpf(lam)
    Cu = [[max(eV * eV * l * l, mpf("0.01") ** 2), mpf(0)], [mpf(0), max(eW * eW * delta * delta, mpf("0.01") ** 2)]]
    Cx = matmul(matmul(dfdu, Cu), transpose(dfdu))
    for i in range(3):
        Cx[i][i] += la * la
    U = cholesky_upper(inv3(Cx))
    A = [[mpf(0)] * 6 for _ in range(6)]
    for i in range(2):
        A[i][0], A[i][1], A[i][5] = U[i][0], U[i][1], U[i][2]
    A[2][2] = A[3][3] = A[4][4] = 1 / la
    A[5][5] = U[2][2]
    return A


def test_odometry_cost_against_50_digits():
    """OdometryCost, src/calibration/odometry_cost_function.cpp: chain of wheel increments :72-94 (odom_zeta_i :10-36,
    zeta_i_jacobian :39-69), calc_acc :96-144, ctor :147-197, Evaluate :202-266 with blocks (6, 6, 3)"""
    for trial in range(4):
        n = int(RNG.integers(1, 7))
        dq = RNG.uniform(0.05, 0.6, (n, 2)) * RNG.choice([1.0, 1.0, -1.0], (n, 1))
        intr0 = np.array([0.11, 0.105, 0.52]) * (1 + 0.02 * RNG.standard_normal(3))
        intr = intr0 * (1 + 0.01 * RNG.standard_normal(3))
        errV, errW, lam = (float(v) for v in RNG.uniform(0.01, 0.2, 3))

        def chain(p):
            r1, r2, g = V(p)
            acc, tfs, jzs = Transf(), [], []
            for dl, dr in dq:
                dl, dr = mpf(float(dl)), mpf(float(dr))
                v = (r1 / 2) * dl + (r2 / 2) * dr
                w = -(r1 / g) * dl + (r2 / g) * dr
                acc = acc.compose(Transf([v, mpf(0), mpf(0)], [mpf(0), mpf(0), w]))
                tfs.append(acc)
                jzs.append([[dl / 2, dr / 2, mpf(0)], [mpf(0)] * 3, [-dl / g, dr / g, (r1 * dl - r2 * dr) / (g * g)]])
            return tfs, jzs

        zp = chain(intr0)[0][-1]
        A = odometry_A(zp, errV, errW, lam)
        blk = vgo.OdometryCost(errV, errW, lam, dq, intr0)
        close(blk.zeta, zp.array(), "zetaPrior of the wheel chain", floor_rel=1e-6)
        close(blk.A, A, "A", floor_rel=1e-6)
        base = np.array([0.4, -0.3, 0.0, 0.0, 0.0, 0.7])
        x1 = base + 0.01 * RNG.standard_normal(6)
        x2 = vgo.compose(base, to_np(zp.array())) + 0.01 * RNG.standard_normal(6)
        X1, X2 = Transf.from_data(x1), Transf.from_data(x2)
        zeta = X1.inverse_compose(X2)
        tfs, jzs = chain(intr)
        zo = tfs[-1]
        delta = zo.inverse_compose(zeta)
        res = matvec(A, delta.array())

        def jblock(X):
            Ri = X.rot_mat_inv()
            RM = matmul(Ri, inter_omega_rot(X.r))
            J = [[mpf(0)] * 6 for _ in range(6)]
            for i in range(3):
                for j in range(3):
                    J[i][j], J[3 + i][3 + j] = Ri[i][j], RM[i][j]
            return J

        J1 = [[-a for a in row] for row in matmul(matmul(A, zeta.screw_transf_inv()), jblock(X1))]
        J2 = matmul(A, jblock(X2))
        ACC = [[mpf(0)] * 3 for _ in range(3)]
        for i in range(n):
            t0j = tfs[i - 1] if i > 0 else Transf()
            tin = tfs[i].inverse().compose(zo)
            Jm = [[1, 0, -tin.t[1]], [0, 1, tin.t[0]], [0, 0, 1]]
            T = matmul(matmul(t0j.rot_mat(), Jm), jzs[i])
            ACC = [[ACC[a][b] + T[a][b] for b in range(3)] for a in range(3)]
        acc63 = [ACC[0], ACC[1], [mpf(0)] * 3, [mpf(0)] * 3, [mpf(0)] * 3, ACC[2]]
        J3 = [[-a for a in row] for row in matmul(matmul(matmul(A, delta.screw_transf_inv()), jblock(zo)), acc63)]
        r, j1, j2, j3 = blk.evaluate(x1, x2, intr)
        close(r, res, "odometry-cost residual", floor_rel=1e-6)
        close(j1, J1, "odometry-cost J1")
        close(j2, J2, "odometry-cost J2")
        close(j3, J3, "odometry-cost J3 (intrinsics)")


@pytest.mark.parametrize("model", ["eucm", "ucm", "mei"])
@pytest.mark.parametrize("L", [1, 2, 3])
def test_restated_jacobians_are_the_derivatives_of_the_restated_projection(model, L):
    """A pin that needs no reference output: the analytic Jacobian blocks of the restatement (intrinsicJacobian,
    projectionJacobian through InterJacobian::dpdxi, both chain directions) must be the derivatives of its own residuals
    with respect to every parameter -- checked at 50 digits with central differences (step 1e-18: truncation and rounding
    both below 1e-30), away from the first-order branches.  A misread term of a Jacobian formula cannot survive this
    unless the same slip sits in the projection; the C oracle is then held to the same formulas at 1e-12 above."""
    grid, obs, status, params = random_case(model, L)
    r0, Ji, Jm = evaluate(model, status, grid, obs, params)
    h = mpf("1e-18")

    def residuals(block, idx, delta):
        pp = [[mpf(float(x)) for x in b] for b in params]
        pp[block][idx] += delta
        # evaluate() converts its inputs with V(): hand it objects that already are mpf
        return evaluate_mp(model, status, grid, obs, pp)[0]

    worst = 0.0
    for block, J in [(0, Ji)] + [(1 + l, Jm[l]) for l in range(L)]:
        ncol = len(params[block])
        for j in range(ncol):
            rp, rm = residuals(block, j, h), residuals(block, j, -h)
            col = [(a - b) / (2 * h) for a, b in zip(rp, rm)]
            ana = [J[i][j] for i in range(len(col))]
            scale = max(max(abs(x) for x in ana), max(abs(x) for x in col), mpf("1e-30"))
            err = max(abs(a - c) for a, c in zip(ana, col)) / scale
            worst = max(worst, float(err))
            assert err < mpf("1e-25"), (model, L, "block %d column %d" % (block, j), float(err))
    print(model, L, "".join("ID"[s] for s in status), "worst |analytic - numeric| / scale = %.1e" % worst)


def reconstruct(model, p, uv):
    """ICamera::reconstructPoint: include/projection/eucm.h:85-106, ucm.h:81-103, mei.h:90-112 (the latter ignores the
    distortion terms, as the reference does) -- the inverse mapping, a different formula from the projector's"""
    if model == "eucm":
        alpha, beta, fu, fv, u0, v0 = p
        xn, yn = (uv[0] - u0) / fu, (uv[1] - v0) / fv
        u2 = xn * xn + yn * yn
        gamma = 1 - alpha
        num = 1 - u2 * alpha * alpha * beta
        det = 1 - (alpha - gamma) * beta * u2
        assert det >= 0
        return [xn, yn, num / (gamma + alpha * mp.sqrt(det))]
    xi = p[0]
    fu, fv, u0, v0 = (p[1], p[2], p[3], p[4]) if model == "ucm" else (p[6], p[7], p[8], p[9])
    xn, yn = (uv[0] - u0) / fu, (uv[1] - v0) / fv
    u2 = xn * xn + yn * yn
    gamma = mp.sqrt(1 + u2 * (1 - xi * xi))
    etanum = -gamma - xi * u2
    etadenom = xi * xi * u2 - 1
    return [xn, yn, etadenom / (etadenom + xi * etanum)]


@pytest.mark.parametrize("model", ["eucm", "ucm", "mei"])
def test_projection_inverts_the_reference_s_own_reconstruction(model):
    """A second pin without reference outputs: the reference carries the inverse of each projection as a separate formula
    (reconstructPoint); restated independently, project(reconstruct(pixel)) must give the pixel back.  Holds to 1e-45 at 50
    digits for EUCM and UCM, and for Mei with its distortion switched off (its reconstructPoint ignores the distortion) --
    so the restated PROJECTIONS are the ones the reference's reconstruction inverts, whatever was read from the sources."""
    p = V(INTR[model])
    if model == "mei":
        p = [p[0]] + [mpf(0)] * 5 + p[6:]
    cam = CAMERAS[model]
    worst = mpf(0)
    for _ in range(40):
        uv = [mpf(float(RNG.uniform(60, 1220))), mpf(float(RNG.uniform(40, 760)))]
        X = reconstruct(model, p, uv)
        s = mpf(float(RNG.uniform(0.3, 4.0)))           # any point of the ray projects to the same pixel
        ok, got, _, _ = cam(p, [s * x for x in X])
        assert ok
        worst = max(worst, abs(got[0] - uv[0]), abs(got[1] - uv[1]))
    assert worst < mpf("1e-42"), float(worst)
    print(model, "worst |project(reconstruct(uv)) - uv| = %.1e px" % float(worst))
