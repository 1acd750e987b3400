"""Independent checks of the CPU oracle: central differences and a float64 autograd formulation
that shares no code (and no intermediate representation: rotation matrices instead of
quaternions) with the oracle.  SURVEY.md section 4: for this path "parity with the reference"
and "mathematically right" coincide (analytic vs central differences <= 5.4e-8)."""
import numpy as np
import pytest
import torch

from oracle import vgo

RNG = np.random.default_rng(20260928)

GT = {
    vgo.MODEL_EUCM: [0.595728, 0.768828, 307.318, 289.542, 642.617, 398.42],
    vgo.MODEL_UCM: [1.47358, 760.18, 716.21, 642.617, 398.42],
    vgo.MODEL_MEI: [1.47358, -0.05, 0.01, -0.002, 0.001, -0.0015, 760.18, 716.21, 642.617, 398.42],
}


def board():
    return np.array([[0.1 * j, 0.1 * i, 0.0] for i in range(8) for j in range(12)])


def random_chain(L):
    """L-long chain whose LAST member puts the board in front of the camera; others are small rig offsets."""
    xis = []
    for l in range(L - 1):
        xis.append(np.concatenate([RNG.uniform(-0.2, 0.2, 3), RNG.uniform(-0.3, 0.3, 3)]))
    xis.append(np.array([-0.55, -0.35, 0.9, 0.0, 0.0, 0.0]) + np.concatenate(
        [RNG.uniform(-0.1, 0.1, 3), RNG.uniform(-0.4, 0.4, 3)]))
    return xis


# ---------------------------------------------------------------- independent torch formulation
def _hat(v):
    z = torch.zeros((), dtype=v.dtype)
    return torch.stack([torch.stack([z, -v[2], v[1]]), torch.stack([v[2], z, -v[0]]),
                        torch.stack([-v[1], v[0], z])])


def _rodrigues(r):
    th = torch.sqrt((r * r).sum())
    K = _hat(r / th)
    return torch.eye(3, dtype=r.dtype) + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)


def _project(model, p, X):
    x, y, z = X[0], X[1], X[2]
    if model == vgo.MODEL_EUCM:
        d = p[0] * torch.sqrt(z * z + p[1] * (x * x + y * y)) + (1 - p[0]) * z
        return torch.stack([p[2] * x / d + p[4], p[3] * y / d + p[5]])
    rho = torch.sqrt(x * x + y * y + z * z)
    xn, yn = x / (z + p[0] * rho), y / (z + p[0] * rho)
    if model == vgo.MODEL_UCM:
        return torch.stack([p[1] * xn + p[3], p[2] * yn + p[4]])
    r2 = xn * xn + yn * yn
    D = 1 + p[1] * r2 + p[2] * r2 ** 2 + p[3] * r2 ** 3
    dx = 2 * p[4] * xn * yn + p[5] * (r2 + 2 * xn * xn)
    dy = 2 * p[5] * xn * yn + p[4] * (r2 + 2 * yn * yn)
    return torch.stack([p[6] * (xn * D + dx) + p[8], p[7] * (yn * D + dy) + p[9]])


def torch_residuals(model, status, grid, obs, params):
    """r(params) with the chain composed as homogeneous matrices (X_cam = T_1^(+-1) ... T_L^(+-1) P)."""
    Racc = torch.eye(3, dtype=torch.float64)
    tacc = torch.zeros(3, dtype=torch.float64)
    for st, xi in zip(status, params[1:]):
        R, t = _rodrigues(xi[3:]), xi[:3]
        if st == vgo.DIRECT:
            tacc = Racc @ t + tacc
            Racc = Racc @ R
        else:
            Racc = Racc @ R.T
            tacc = tacc - Racc @ t
    out = []
    for P, o in zip(grid, obs):
        X = Racc @ torch.tensor(P) + tacc
        out.append(_project(model, params[0], X) - torch.tensor(o))
    return torch.cat(out)


CASES = [(m, st) for m in (vgo.MODEL_EUCM, vgo.MODEL_UCM, vgo.MODEL_MEI)
         for st in ([0], [1, 0], [0, 1, 0], [1, 0, 0, 1, 0])]


@pytest.mark.parametrize("model,status", CASES)
def test_autograd_formulation_matches_oracle(model, status):
    grid = board()[::7]  # 14 corners are plenty
    xis = random_chain(len(status))
    intr = np.array(GT[model])
    proj, _ = vgo.eval_block(model, status, grid, np.zeros((len(grid), 2)), [intr] + xis, want_jac=False)
    obs = proj.reshape(-1, 2) + RNG.normal(0, 0.5, (len(grid), 2))
    res, J = vgo.eval_block(model, status, grid, obs, [intr] + xis)
    assert np.all(np.abs(res) < 1e3)

    tp = [torch.tensor(intr, dtype=torch.float64)] + [torch.tensor(x, dtype=torch.float64) for x in xis]
    f = lambda *p: torch_residuals(model, status, grid, obs, list(p))
    r_t = f(*tp).numpy()
    J_t = torch.autograd.functional.jacobian(f, tuple(tp))
    assert np.max(np.abs(r_t - res)) < 1e-9
    for Jo, Jt in zip(J, J_t):
        Jt = Jt.numpy()
        scale = np.max(np.abs(Jt))
        assert np.max(np.abs(Jo - Jt)) < 1e-9 * scale


@pytest.mark.parametrize("model,status", CASES[::2])
def test_central_differences(model, status):
    grid = board()[::11]
    xis = random_chain(len(status))
    intr = np.array(GT[model])
    obs = np.zeros((len(grid), 2))
    params = [intr] + xis
    _, J = vgo.eval_block(model, status, grid, obs, params)
    for b, p in enumerate(params):
        for c in range(p.size):
            h = 1e-6 * max(1.0, abs(p[c]))
            pp = [q.copy() for q in params]
            pm = [q.copy() for q in params]
            pp[b][c] += h
            pm[b][c] -= h
            rp, _ = vgo.eval_block(model, status, grid, obs, pp, want_jac=False)
            rm, _ = vgo.eval_block(model, status, grid, obs, pm, want_jac=False)
            fd = (rp - rm) / (2 * h)
            scale = max(np.max(np.abs(J[b][:, c])), 1e-3 * np.max(np.abs(J[b])))
            assert np.max(np.abs(fd - J[b][:, c])) < 2e-6 * scale


# ---------------------------------------------------------------- branch behaviour of the geometry
def test_quaternion_small_angle_branch_is_first_order_and_unnormalised():
    # quaternion.h:34-40: |theta| < 1e-6 -> (r/2, 1), NOT normalised
    r = np.array([3e-7, -4e-7, 1e-7])
    q = vgo.quat_from_rotvec(r)
    assert np.array_equal(q, np.array([r[0] / 2, r[1] / 2, r[2] / 2, 1.0]))
    r = np.array([3e-6, -4e-6, 1e-6])  # above the threshold: exact formula
    q = vgo.quat_from_rotvec(r)
    assert abs(np.linalg.norm(q) - 1) < 1e-15 and q[3] < 1.0


def test_to_rotation_vector_branches():
    # quaternion.h:88-91: s < 1e-5 -> 2*(x,y,z)
    assert np.array_equal(vgo.quat_to_rotvec([1e-6, -2e-6, 3e-6, 1.0]), 2 * np.array([1e-6, -2e-6, 3e-6]))
    # |rot| > pi renormalised into (-pi, pi] by normalizeAngle (geometry_core.h:32-38)
    r = np.array([0.0, 0.0, 3.5])
    back = vgo.compose(np.zeros(6), np.concatenate([[0, 0, 0], r]))[3:]
    assert abs(back[2] - (3.5 - 2 * np.pi)) < 1e-14 and np.all(back[:2] == 0)


def test_rotation_matrix_and_interaction_small_angle_branches():
    v = np.array([3e-6, -4e-6, 5e-6])  # |v| < 1e-5 -> I + hat(v)  (geometry_core.h:45-52)
    R = vgo.rotation_matrix(v)
    assert np.array_equal(R, np.array([[1, -v[2], v[1]], [v[2], 1, -v[0]], [-v[1], v[0], 1]]))
    M = vgo.inter_omega_rot(v)  # I + hat(v/2)  (geometry_core.h:163-169)
    h = v / 2
    assert np.array_equal(M, np.array([[1, -h[2], h[1]], [h[2], 1, -h[0]], [-h[1], h[0], 1]]))
    v = np.array([0.3, -0.4, 0.1])
    R = vgo.rotation_matrix(v)
    assert np.max(np.abs(R @ R.T - np.eye(3))) < 1e-15
    assert np.max(np.abs(R - _rodrigues(torch.tensor(v)).numpy())) < 1e-15


def test_compose_inverse_undoes_compose():
    a = np.array([0.1, -0.2, 0.3, 0.2, -0.1, 0.4])
    b = np.array([-0.3, 0.1, 0.5, -0.5, 0.3, 0.2])
    ab = vgo.compose(a, b)
    back = vgo.compose(ab, b, inverse=True)
    assert np.max(np.abs(back - a)) < 1e-15


def test_chain_length_zero_is_legal():
    # intrinsics-only block, camera frame = board frame
    grid = board()[::13] + np.array([-0.5, -0.3, 1.0])
    res, J = vgo.eval_block(vgo.MODEL_UCM, [], grid, np.zeros((len(grid), 2)), [GT[vgo.MODEL_UCM]])
    assert len(J) == 1 and J[0].shape == (2 * len(grid), 5) and np.all(np.isfinite(res))


def test_block_gram_matches_numpy():
    grid = board()
    xis = random_chain(2)
    res, J = vgo.eval_block(vgo.MODEL_MEI, [1, 0], grid, np.full((96, 2), 500.0), [GT[vgo.MODEL_MEI]] + xis)
    S = np.concatenate([J[0], J[1], J[2], res[:, None]], axis=1)
    ref = (S.astype(np.longdouble).T @ S.astype(np.longdouble)).astype(np.float64)
    g = vgo.block_gram(res, J[0], J[1:])
    gf = vgo.block_gram(res, J[0], J[1:], fast=True)
    n = np.linalg.norm(ref)
    assert np.linalg.norm(g - ref) < 1e-15 * n
    assert np.linalg.norm(gf - ref) < 1e-13 * n


def test_eval_dataset_equals_block_by_block():
    grid = board()
    nb = 5
    intr = np.array(GT[vgo.MODEL_EUCM])
    glob = random_chain(1)[0] * 0.1
    poses = np.stack([random_chain(1)[0] for _ in range(nb)])
    pv = np.concatenate([intr, glob, poses.reshape(-1)])
    obs = RNG.uniform(100, 900, (nb, 192))
    res, ji, jm = vgo.eval_dataset(vgo.MODEL_EUCM, [1, 0], grid, obs, pv, 0, [6, 12], [0, 6],
                                   np.arange(nb), threads=2)
    for b in range(nb):
        r1, J1 = vgo.eval_block(vgo.MODEL_EUCM, [1, 0], grid, obs[b].reshape(-1, 2), [intr, glob, poses[b]])
        assert np.array_equal(r1, res[b]) and np.array_equal(J1[0], ji[b])
        assert np.array_equal(J1[1], jm[0][b]) and np.array_equal(J1[2], jm[1][b])


def test_transformation_prior_block():
    """TransformationPrior (calib_cost_functions.h:79-103, .cpp:214-228): zero at the prior, A-weighted and rotated
    relative transform elsewhere, constant Jacobian A = diag(stiffness) with the rotation block times interOmegaRot."""
    st = np.array([10.0, 20.0, 30.0, 4.0, 5.0, 6.0])
    xp = np.array([0.2, -0.1, 0.05, 0.3, -0.4, 0.1])
    r0, J = vgo.transformation_prior(st, xp, xp)
    assert np.max(np.abs(r0)) < 1e-15
    M = vgo.inter_omega_rot(xp[3:])
    A = np.zeros((6, 6))
    A[:3, :3] = np.diag(st[:3])
    A[3:, 3:] = np.diag(st[3:]) @ M
    assert np.max(np.abs(J - A)) < 1e-15
    x = xp + np.array([0.01, -0.02, 0.005, 0.002, 0.001, -0.003])
    r, _ = vgo.transformation_prior(st, xp, x)
    # independent: e = prior^-1 o xi through matrices
    Rp, Rx = _rodrigues(torch.tensor(xp[3:])).numpy(), _rodrigues(torch.tensor(x[3:])).numpy()
    et = Rp.T @ (x[:3] - xp[:3])
    Re = Rp.T @ Rx
    ang = np.arccos((np.trace(Re) - 1) / 2)
    er = ang / (2 * np.sin(ang)) * np.array([Re[2, 1] - Re[1, 2], Re[0, 2] - Re[2, 0], Re[1, 0] - Re[0, 1]])
    ref = A @ np.concatenate([Rp @ et, Rp @ er])
    assert np.max(np.abs(r - ref)) < 1e-12


def _se3(xi):
    T = np.eye(4)
    T[:3, :3] = _rodrigues(torch.tensor(np.asarray(xi[3:], dtype=np.float64))).numpy()
    T[:3, 3] = xi[:3]
    return T


def _log_rot(R):
    ang = np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
    if ang < 1e-12:
        return np.zeros(3)
    return ang / (2 * np.sin(ang)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])


def test_odometry_prior_block():
    """OdometryPrior (calib_cost_functions.h:64-77, .cpp:119-212): r = A (zetaPrior^-1 o xi1^-1 o xi2).  Zero when the
    pair reproduces the measured increment (whatever the absolute pose), independent matrix formulation elsewhere, the
    structure of A (upper-triangular planar block in (x, y, yaw), 1/lambda on z, roll, pitch), and the known quality of
    the reference's analytic Jacobians: right-multiplied increments are treated as additive, which is first-order
    accurate only for small rotation vectors."""
    rng = np.random.default_rng(5)
    o1 = np.array([0.3, -0.2, 0.0, 0.0, 0.0, 0.4])
    o2 = np.array([0.55, -0.05, 0.0, 0.0, 0.0, 0.65])
    errV, errW, lam = 0.05, 0.02, 0.3
    blk = vgo.OdometryPrior(errV, errW, lam, o1, o2)
    # zeta = o1^-1 o o2
    Z = np.linalg.inv(_se3(o1)) @ _se3(o2)
    assert np.max(np.abs(blk.zeta[:3] - Z[:3, 3])) < 1e-14 and np.max(np.abs(blk.zeta[3:] - _log_rot(Z[:3, :3]))) < 1e-14
    # A: Cholesky factor (upper) of the inverse planar covariance, spread over (x, y, yaw); 1/lambda elsewhere
    delta, l = max(np.linalg.norm(blk.zeta[3:]), 0.01), max(np.linalg.norm(blk.zeta[:3]), 0.01)
    s, c = np.sin(delta / 2), np.cos(delta / 2)
    dfdu = np.array([[c, l / 2 * s], [-s, l / 2 * c], [0, 1]])
    Cu = np.diag([max(errV**2 * l**2, 1e-4), max(errW**2 * delta**2, 1e-4)])
    Cx = dfdu @ Cu @ dfdu.T + lam**2 * np.eye(3)
    U = np.linalg.cholesky(np.linalg.inv(Cx)).T
    A = np.zeros((6, 6))
    A[:2, :2] = U[:2, :2]
    A[:2, 5] = U[:2, 2]
    A[2, 2] = A[3, 3] = A[4, 4] = 1 / lam
    A[5, 5] = U[2, 2]
    assert np.max(np.abs(blk.A - A)) < 1e-11 * np.max(np.abs(A))
    # zero for ANY absolute pose that reproduces the increment
    base = np.array([1.0, 2.0, -0.5, 0.2, -0.3, 0.9])
    x1 = vgo.compose(base, o1)
    x2 = vgo.compose(base, o2)
    r0, _, _ = blk.evaluate(x1, x2)
    assert np.max(np.abs(r0)) < 1e-12
    # independent formulation away from the prior
    x1 = o1 + 0.02 * rng.standard_normal(6)
    x2 = o2 + 0.02 * rng.standard_normal(6)
    r, J1, J2 = blk.evaluate(x1, x2)
    E = np.linalg.inv(Z) @ np.linalg.inv(_se3(x1)) @ _se3(x2)
    ref = A @ np.concatenate([E[:3, 3], _log_rot(E[:3, :3])])
    assert np.max(np.abs(r - ref)) < 1e-12
    # Jacobians against central differences: a few percent, not 1e-7 (documented reference behaviour)
    def fd(which):
        J = np.zeros((6, 6))
        for k in range(6):
            d = np.zeros(6)
            d[k] = 1e-6
            a = blk.evaluate(x1 + d, x2)[0] if which == 0 else blk.evaluate(x1, x2 + d)[0]
            b = blk.evaluate(x1 - d, x2)[0] if which == 0 else blk.evaluate(x1, x2 - d)[0]
            J[:, k] = (a - b) / 2e-6
        return J
    for J, F in ((J1, fd(0)), (J2, fd(1))):
        assert np.max(np.abs(J - F)) < 0.05 * np.max(np.abs(F))
        assert np.max(np.abs(J - F)) > 1e-6 * np.max(np.abs(F))   # ... and they really are inexact
