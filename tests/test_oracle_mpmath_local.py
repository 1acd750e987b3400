"""SURVEY 8(f) rank 5 -- the localization costs on the same camera models -- in the oracle (oracle/vg_oracle.c, double)
against a second, independent restatement in 50-digit mpmath, written from the reference sources:
  CameraJacobian        include/projection/jacobian.h:51-119
  Triangulator          src/reconstruction/triangulator.cpp:114-259 (computeRegular, regDiv)
  MonoReprojectCost     src/localization/local_cost_functions.cpp:216-278
  SparseReprojectCost   src/localization/local_cost_functions.cpp:281-391
Same rules as tests/test_oracle_mpmath.py (whose geometry / camera restatements are reused): 1e-12, test points away
from branch thresholds.  Also here: which of these Jacobians are exact derivatives of their residuals (central differences
of the 50-digit restatement) -- MonoReprojectCost's are; SparseReprojectCost's are only when the base->camera rotation is
the identity, and its v-rows are not divided by the feature size although the residuals are (both reproduced as written).
"""
import numpy as np
import pytest

mp = pytest.importorskip("mpmath")
from mpmath import mpf  # noqa: E402

from oracle import vgo  # noqa: E402
from tests import test_oracle_mpmath as M  # noqa: E402

mp.mp.dps = 50
V, Transf, matmul, matvec, hat, transpose = M.V, M.Transf, M.matmul, M.matvec, M.hat, M.transpose
MODELS = {"eucm": 0, "ucm": 1, "mei": 2}
BIG = M.BIG


def close_ill(got, ref_mp, what, norm_tol=1e-12, elem_tol=1e-10):
    """M.close with separate tolerances: the triangulated depth is a ratio of two differences of near-equal products
    (condition number ~ depth / baseline, 10-100 in the scenes below), so single entries of a double evaluation carry a few
    1e-12 of the block's scale; the block as a whole is still held to 1e-12"""
    got = np.asarray(got, float)
    ref = M.to_np(ref_mp)
    flat = [x for r in ref_mp for x in r] if isinstance(ref_mp[0], list) else list(ref_mp)
    diff = np.array([float(mpf(float(g)) - r) for g, r in zip(got.ravel(), flat)]).reshape(got.shape)
    e1 = np.linalg.norm(diff) / np.linalg.norm(ref)
    e2 = np.max(np.abs(diff) / np.maximum(np.abs(ref), 1e-3 * np.max(np.abs(ref))))
    assert e1 <= norm_tol and e2 <= elem_tol, "%s: normwise %.2e elementwise %.2e" % (what, e1, e2)


def neg(A):
    return [[-a for a in r] for r in A]


def camera_jacobian_mp(model, intr, T12, T23, X2, grad=None):
    """CameraJacobian ctor (jacobian.h:54-71) + dpdxi (:75-96) + dfdxi (:99-113)"""
    R21 = T12.rot_mat_inv()
    Mo = M.inter_omega_rot(T12.r)
    if T23 is not None:
        R32 = T23.rot_mat_inv()
        L11 = matmul(R32, R21)
        L22 = matmul(L11, Mo)
        L12 = matmul(matmul(matmul(neg(R32), hat(T23.t)), R21), Mo)
    else:
        L11 = R21
        L12 = [[mpf(0)] * 3 for _ in range(3)]
        L22 = matmul(L11, Mo)
    ok, _, P, _ = M.CAMERAS[model](intr, X2)
    if not ok:
        z = [mpf(0)] * 6
        return z, z, z
    B = matmul(hat(X2), L22)
    if T23 is not None:
        B = [[B[i][j] - L12[i][j] for j in range(3)] for i in range(3)]
    rows = []
    for row in range(2):
        rows.append(matmul([[-a for a in P[row]]], L11)[0] + matmul([P[row]], B)[0])
    df = None
    if grad is not None:
        d = [grad[0] * P[0][j] + grad[1] * P[1][j] for j in range(3)]
        df = matmul([[-a for a in d]], L11)[0] + matmul([d], B)[0]
    return rows[0], rows[1], df


def reg_div(num, den, eps):  # triangulator.cpp:114-128
    if den > eps * num:
        return num / den
    if num == 0:
        return 2 / eps
    return 2 / eps - den / (num * eps * eps)


def dot(a, b):
    return sum(x * y for x, y in zip(a, b))


def triangulate_regular_mp(xi, eps, p, q):
    """Triangulator(xi, eps).computeRegular(p, q, res1, res2, jac1, jac2), triangulator.cpp:145-259"""
    R, t = xi.rot_mat(), xi.t
    q = matvec(R, q)
    r = [a + b for a, b in zip(p, q)]
    tp, tq, tr, tt, rp, rq = dot(t, p), dot(t, q), dot(t, r), dot(t, t), dot(r, p), dot(r, q)
    delta = tp * rq - tq * rp
    d1, d2 = tt * rq - tr * tq, tt * rp - tr * tp
    l1, l2 = reg_div(d1, delta, eps), reg_div(d2, delta, eps)
    qs = hat(q)
    dV = [rq * a - rp * b for a, b in zip(p, q)]
    dO = matvec(qs, [tp * a - tq * a - rp * c for a, c in zip(p, t)])
    d1V = [2 * rq * c - tq * a - tr * b for a, b, c in zip(r, q, t)]
    d1O = matvec(qs, [tt * a - tq * c - tr * c for a, c in zip(p, t)])
    d2V = [2 * rp * c - tp * a - tr * b for a, b, c in zip(r, p, t)]
    d2O = matvec(qs, [tt * a - tp * c for a, c in zip(p, t)])
    out = []
    for lam, dd, dv, do, sign in ((l1, d1, d1V, d1O, 1), (l2, d2, d2V, d2O, -1)):
        if delta > eps * dd:
            j = [(a - lam * b) / delta for a, b in zip(dv, dV)] + [(a - lam * b) / delta for a, b in zip(do, dO)]
        elif dd == 0:
            j = [mpf(0)] * 6
        elif sign == 1:   # :209-217
            coef = 1 / (eps * eps)
            k, k1 = -coef / dd, coef * delta / dd / dd
            j = [k * b + k1 * a for a, b in zip(dv, dV)] + [k * b + k1 * a for a, b in zip(do, dO)]
        else:             # :238-246
            coef = -1 / (eps * eps)
            k, k2 = coef / dd, coef * delta / dd / dd
            j = [k * b - k2 * a for a, b in zip(dv, dV)] + [k * b - k2 * a for a, b in zip(do, dO)]
        out.append(j)
    return l1, l2, out[0], out[1]


def inter_jacobian_mp(xi13, xi23, inverted):
    """InterJacobian ctor, jacobian.h:139-152"""
    R12 = matmul(xi13.rot_mat(), xi23.rot_mat_inv())
    M12 = matmul(R12, M.inter_omega_rot(xi23.r))
    if inverted:
        R12, M12 = neg(R12), neg(M12)
    return R12, M12, list(xi13.t)


def dpdxi_mp(frame, P, X):
    """InterJacobian::dpdxi, jacobian.h:155-171"""
    R12, M12, t13 = frame
    H = hat([a - b for a, b in zip(X, t13)])
    return [matmul([P[row]], R12)[0] + matmul(matmul([[-a for a in P[row]]], H), M12)[0] for row in range(2)]


def mono_reproject_mp(model, intr, xb, x1, p2, xo, lengths):
    """MonoReprojectCost::Evaluate, local_cost_functions.cpp:216-278"""
    cam = M.CAMERAS[model]
    xi21 = xb.inverse_compose(xo.inverse_compose(xb))
    R21 = xi21.rot_mat()
    xv2 = [[a + b for a, b in zip(matvec(R21, [c * lengths[i] for c in x1[i]]), xi21.t)] for i in range(5)]
    res, J0 = [], []
    J1 = [[mpf(0)] * 5 for _ in range(10)]
    frame = inter_jacobian_mp(xb.inverse(), xo, True)
    for i in range(5):
        ok, uv, P, _ = cam(intr, xv2[i])
        res += [uv[0] - p2[i][0], uv[1] - p2[i][1]] if ok else [BIG, BIG]
        J0 += dpdxi_mp(frame, P, xv2[i])
        n2 = matvec(R21, x1[i])
        J1[2 * i][i], J1[2 * i + 1][i] = dot(P[0], n2), dot(P[1], n2)
    return res, J0, J1


def sparse_reproject_mp(model, intr, xb, x1, x2, p2, size, xo, fix_v_rows=False):
    """SparseReprojectCost::Evaluate, local_cost_functions.cpp:281-391.  fix_v_rows: divide the v-rows by the size too
    (what the derivative of the residual is; the reference divides only [i*12, i*12 + 6), :383-389)"""
    cam = M.CAMERAS[model]
    n = len(x1)
    xi12 = xb.inverse_compose(xo.compose(xb))
    eps = mpf("1e-3")  # triangulator.h:34
    tri = [triangulate_regular_mp(xi12, eps, x1[i], x2[i]) for i in range(n)]
    R21 = xi12.rot_mat_inv()
    xv2 = [matvec(R21, [c * tri[i][0] - t for c, t in zip(x1[i], xi12.t)]) for i in range(n)]
    frame = inter_jacobian_mp(xb.inverse(), xo, True)
    Rcb = xb.rot_mat_inv()
    Mm = matmul(Rcb, M.inter_omega_rot(xo.r))
    tbc1 = matvec(matmul(Rcb, transpose(R21)), xb.t)
    Q = matmul(neg(hat(tbc1)), Mm)
    res, J = [], []
    for i in range(n):
        ok, uv, P, _ = cam(intr, xv2[i])
        if not ok:
            res += [BIG, BIG]
            J += [[mpf(0)] * 6, [mpf(0)] * 6]
            continue
        res += [(uv[0] - p2[i][0]) / size[i], (uv[1] - p2[i][1]) / size[i]]
        rows = dpdxi_mp(frame, P, xv2[i])
        n2 = matvec(R21, x1[i])
        dpdl = [dot(P[0], n2), dot(P[1], n2)]
        dldv, dldw = tri[i][2][:3], tri[i][2][3:]
        dldt = matmul([dldv], Rcb)[0]
        dldr = [a + b for a, b in zip(matmul([dldw], Mm)[0], matmul([dldv], Q)[0])]
        for row in range(2):
            full = [rows[row][j] + dpdl[row] * dldt[j] for j in range(3)] + [rows[row][3 + j] + dpdl[row] * dldr[j] for j in range(3)]
            if row == 0 or fix_v_rows:
                full = [a / size[i] for a in full]
            J.append(full)
    return res, J


# ------------------------------------------------------------------------------------------ cases
RNG = np.random.default_rng(20260930)
XB = np.array([0.21, -0.08, 0.33, 0.12, -1.15, 1.07])      # base -> camera, a camera looking sideways
XO = np.array([0.31, 0.04, -0.02, 0.012, -0.021, 0.083])   # odometry increment


def scene(n, xb, xo, noise=1e-3):
    """n points in front of camera 1, unit direction vectors in both frames (the second ones slightly inconsistent, as
    matched key points are), observations and feature sizes"""
    X1 = np.column_stack([RNG.uniform(-1, 1, n), RNG.uniform(-0.7, 0.7, n), RNG.uniform(2, 6, n)])
    x1 = X1 / np.linalg.norm(X1, axis=1)[:, None]

    def T(x):
        A = np.eye(4)
        A[:3, :3], A[:3, 3] = vgo.rotation_matrix(x[3:]), x[:3]
        return A

    T12 = np.linalg.inv(T(xb)) @ T(xo) @ T(xb)
    X2 = (np.linalg.inv(T12) @ np.c_[X1, np.ones(n)].T).T[:, :3]
    x2 = X2 / np.linalg.norm(X2, axis=1)[:, None] + noise * RNG.standard_normal((n, 3))
    return X1, x1, x2, RNG.uniform(300, 900, (n, 2)), RNG.uniform(1, 4, n)


@pytest.mark.parametrize("model", ["eucm", "ucm", "mei"])
@pytest.mark.parametrize("two", [False, True])
def test_camera_jacobian_against_50_digits(model, two):
    T12 = np.array([0.3, -0.2, 0.1, 0.4, -0.3, 0.2])
    T23 = np.array([-0.1, 0.25, 0.05, -0.2, 0.1, 0.5]) if two else None
    X = np.column_stack([RNG.uniform(-1, 1, 6), RNG.uniform(-0.7, 0.7, 6), RNG.uniform(1.5, 5, 6)])
    grad = RNG.standard_normal((6, 2))
    du, dv, df = vgo.camera_jacobian(MODELS[model], M.INTR[model], T12, T23, X, grad)
    for i in range(6):
        ru, rv, rf = camera_jacobian_mp(model, V(M.INTR[model]), Transf.from_data(T12), Transf.from_data(T23) if two else None,
                                        V(X[i]), V(grad[i]))
        M.close(du[i], ru, "dudxi")
        M.close(dv[i], rv, "dvdxi")
        M.close(df[i], rf, "dfdxi")


def test_camera_jacobian_of_a_failed_projection_is_zero():
    du, dv, df = vgo.camera_jacobian(0, M.INTR["eucm"], [0, 0, 0, 0.1, 0, 0], None, [[0.1, 0.2, -3.0]], [[1.0, 2.0]])
    assert not du.any() and not dv.any() and not df.any()
    ru, rv, rf = camera_jacobian_mp("eucm", V(M.INTR["eucm"]), Transf.from_data([0, 0, 0, 0.1, 0, 0]), None, V([0.1, 0.2, -3.0]), V([1, 2]))
    assert not any(ru) and not any(rv) and not any(rf)


@pytest.mark.parametrize("case", ["regular", "near_parallel", "behind"])
def test_triangulator_against_50_digits(case):
    """regular: delta > eps * delta1 (both branches 1); near_parallel: a far point, the regularised branch and its
    Jacobian (:209-217, :238-246); behind: negative delta"""
    xi = np.array([0.3, -0.05, 0.1, 0.05, -0.1, 0.2])
    R, t = vgo.rotation_matrix(xi[3:]), xi[:3]
    X1 = {"regular": [0.4, -0.3, 3.0], "near_parallel": [130.0, -100.0, 3000.0], "behind": [0.4, -0.3, 3.0]}[case]
    X1 = np.array(X1)
    p = X1 / np.linalg.norm(X1)
    X2 = R.T @ (X1 - t)
    q = X2 / np.linalg.norm(X2) + (1e-6 if case == "near_parallel" else 1e-4) * RNG.standard_normal(3)
    if case == "behind":
        q = -q
    l1, l2, j1, j2 = vgo.triangulate_regular(xi, p, q, 1e-3)
    m1, m2, mj1, mj2 = triangulate_regular_mp(Transf.from_data(xi), mpf("1e-3"), V(p), V(q))
    if case == "near_parallel":   # the regularised branch is really taken
        qq = R @ q
        r = p + qq
        delta = (t @ p) * (r @ qq) - (t @ qq) * (r @ p)
        assert not delta > 1e-3 * ((t @ t) * (r @ qq) - (t @ r) * (t @ qq))
    assert abs(mpf(l1) - m1) <= 1e-12 * abs(m1) and abs(mpf(l2) - m2) <= 1e-12 * abs(m2)
    M.close(j1, mj1, "jac1")
    M.close(j2, mj2, "jac2")
    if case == "regular":
        assert abs(l1 - np.linalg.norm(X1)) < 1e-2 and abs(l2 - np.linalg.norm(X2)) < 1e-2


@pytest.mark.parametrize("model", ["eucm", "ucm", "mei"])
def test_mono_reproject_against_50_digits(model):
    X1, x1, _, p2, _ = scene(5, XB, XO)
    lengths = np.linalg.norm(X1, axis=1) * (1 + 0.01 * RNG.standard_normal(5))
    r, j0, j1 = vgo.mono_reproject(MODELS[model], M.INTR[model], XB, x1, p2, XO, lengths)
    mr, m0, m1 = mono_reproject_mp(model, V(M.INTR[model]), Transf.from_data(XB), [V(a) for a in x1], [V(a) for a in p2],
                                   Transf.from_data(XO), V(lengths))
    # a residual is a difference of near-equal numbers: absolute tolerance relative to the projection
    for k in range(10):
        assert abs(mpf(float(r[k])) - mr[k]) <= 1e-12 * 1e3
    M.close(j0, m0, "mono odometry jacobian")
    M.close(j1, m1, "mono length jacobian")
    assert np.count_nonzero(j1) == 10


@pytest.mark.parametrize("model", ["eucm", "ucm", "mei"])
def test_sparse_reproject_against_50_digits(model):
    _, x1, x2, p2, size = scene(9, XB, XO)
    r, J = vgo.sparse_reproject(MODELS[model], M.INTR[model], XB, x1, x2, p2, size, XO)
    mr, mJ = sparse_reproject_mp(model, V(M.INTR[model]), Transf.from_data(XB), [V(a) for a in x1], [V(a) for a in x2],
                                 [V(a) for a in p2], V(size), Transf.from_data(XO))
    for k in range(18):
        assert abs(mpf(float(r[k])) - mr[k]) <= 1e-11 * 1e3
    close_ill(J, mJ, "sparse jacobian")


def test_failed_projections_in_both_costs():
    """EUCM points that end up behind the camera: residual pair 1e15; MonoReprojectCost keeps whatever dpdxi gives
    (zero rows: the camera zeroes its Jacobian), SparseReprojectCost zeroes the rows explicitly and skips them (:336, :362)"""
    intr = M.INTR["eucm"]
    X1, x1, x2, p2, size = scene(6, XB, XO)
    x1b, x2b = x1.copy(), x2.copy()
    x1b[2], x2b[2] = -x1[2], -x2[2]           # a pair of directions that triangulates behind both cameras
    r, J = vgo.sparse_reproject(0, intr, XB, x1b, x2b, p2, size, XO)
    assert r[4] == 1e15 and r[5] == 1e15 and not J[4:6].any() and np.all(np.abs(r[:4]) < 1e4) and J[:4].any()
    mr, mJ = sparse_reproject_mp("eucm", V(intr), Transf.from_data(XB), [V(a) for a in x1b], [V(a) for a in x2b],
                                 [V(a) for a in p2], V(size), Transf.from_data(XO))
    assert mr[4] == BIG and not any(mJ[4]) and not any(mJ[5])
    close_ill(J, mJ, "sparse jacobian with a failed point")
    lengths = np.linalg.norm(X1, axis=1)[:5].copy()
    lengths[1] = -lengths[1]                  # negative length: the point is behind the camera
    r, j0, j1 = vgo.mono_reproject(0, intr, XB, x1[:5], p2[:5], XO, lengths)
    assert r[2] == 1e15 and r[3] == 1e15 and not j0[2:4].any() and j1[2, 1] == 0 and j1[3, 1] == 0 and j0[:2].any()


@pytest.mark.parametrize("model", ["eucm", "mei"])
def test_which_of_these_jacobians_are_derivatives(model):
    """central differences of the 50-digit restatement, step 1e-20: MonoReprojectCost's two blocks are exact derivatives;
    SparseReprojectCost's rows (v-rows divided by the size) are exact when the base->camera rotation is the identity and
    off by up to a percent otherwise (tBaseCam1 = RcamBase * R21^T * t, :355, conjugates the odometry rotation once too
    often) -- a property of the reference, kept as it is."""
    h = mpf("1e-20")
    intr = V(M.INTR[model])
    X1, x1, x2, p2, size = scene(5, XB, XO)
    lengths = np.linalg.norm(X1, axis=1)
    a1, a2, ap, asz, al = [V(a) for a in x1], [V(a) for a in x2], [V(a) for a in p2], V(size), V(lengths)
    xb, xo = Transf.from_data(XB), V(XO)

    def tf(v):
        return Transf(list(v[:3]), list(v[3:]))

    _, m0, m1 = mono_reproject_mp(model, intr, xb, a1, ap, tf(xo), al)
    for k in range(6):
        up = mono_reproject_mp(model, intr, xb, a1, ap, tf([v + (h if j == k else 0) for j, v in enumerate(xo)]), al)[0]
        dn = mono_reproject_mp(model, intr, xb, a1, ap, tf([v - (h if j == k else 0) for j, v in enumerate(xo)]), al)[0]
        for row in range(10):
            assert abs((up[row] - dn[row]) / (2 * h) - m0[row][k]) <= mpf("1e-25") * max(1, abs(m0[row][k]))
    for k in range(5):
        up = mono_reproject_mp(model, intr, xb, a1, ap, tf(xo), [v + (h if j == k else 0) for j, v in enumerate(al)])[0]
        dn = mono_reproject_mp(model, intr, xb, a1, ap, tf(xo), [v - (h if j == k else 0) for j, v in enumerate(al)])[0]
        for row in range(10):
            assert abs((up[row] - dn[row]) / (2 * h) - m1[row][k]) <= mpf("1e-25") * max(1, abs(m1[row][k]))

    def sparse_err(xbv):
        xbt = Transf.from_data(xbv)
        _, mJ = sparse_reproject_mp(model, intr, xbt, a1, a2, ap, asz, tf(xo), fix_v_rows=True)
        worst, scale = mpf(0), max(abs(v) for r in mJ for v in r)
        for k in range(6):
            up = sparse_reproject_mp(model, intr, xbt, a1, a2, ap, asz, tf([v + (h if j == k else 0) for j, v in enumerate(xo)]))[0]
            dn = sparse_reproject_mp(model, intr, xbt, a1, a2, ap, asz, tf([v - (h if j == k else 0) for j, v in enumerate(xo)]))[0]
            for row in range(10):
                worst = max(worst, abs((up[row] - dn[row]) / (2 * h) - mJ[row][k]) / scale)
        return worst

    assert sparse_err([0.21, -0.08, 0.33, 0, 0, 0]) <= mpf("1e-25")
    assert mpf("1e-4") < sparse_err(XB) < mpf("5e-2")
