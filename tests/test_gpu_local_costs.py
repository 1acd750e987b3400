"""SURVEY 8(f) rank 5 on the GPU: MonoReprojectCost / SparseReprojectCost (src/localization/local_cost_functions.cpp:216-391),
Triangulator::computeRegular inside the latter, CameraJacobian (include/projection/jacobian.h:51-119) -- HIP kernels behind the
C ABI (include/visgeom_amd.h section 6) against the oracle's restatements (oracle/vg_oracle.c, held to 50-digit mpmath by
tests/test_oracle_mpmath_local.py).  Bar: north_star's 1e-10 with the SURVEY 8(c) metric; failed projections (1e15 pair,
zero rows) must match exactly."""
import numpy as np
import pytest

from oracle import vgo
from tests.parity import BIG, TOL, assert_block_parity

pytestmark = pytest.mark.gpu

MODELS = {"eucm": 0, "ucm": 1, "mei": 2}
INTR = {"eucm": [0.571, 1.18, 312.0, 305.0, 655.0, 391.0], "ucm": [1.31, 702.0, 694.0, 633.0, 409.0],
        "mei": [1.27, -0.04, 0.012, -0.003, 0.0012, -0.0017, 698.0, 705.0, 648.0, 395.0]}
XB = np.array([0.21, -0.08, 0.33, 0.12, -1.15, 1.07])


@pytest.fixture(scope="module")
def loc():
    import torch

    assert torch.cuda.is_available()
    from visgeom_amd import localization

    return localization


def T(x):
    A = np.eye(4)
    A[:3, :3], A[:3, 3] = vgo.rotation_matrix(x[3:]), x[:3]
    return A


def scene(rng, n, xb, xo, noise=1e-3, far=0):
    """n points in front of camera 1 (the last `far` of them hundreds of metres away: the regularised triangulation),
    unit directions in both camera frames, observations, feature sizes"""
    X1 = np.column_stack([rng.uniform(-1, 1, n), rng.uniform(-0.7, 0.7, n), rng.uniform(2, 6, n)])
    if far:
        X1[n - far:] *= rng.uniform(300, 900, (far, 1))
    x1 = X1 / np.linalg.norm(X1, axis=1)[:, None]
    T12 = np.linalg.inv(T(xb)) @ T(xo) @ T(xb)
    X2 = (np.linalg.inv(T12) @ np.c_[X1, np.ones(n)].T).T[:, :3]
    sigma = np.full((n, 1), noise)
    if far:
        sigma[n - far:] = 1e-7
    x2 = X2 / np.linalg.norm(X2, axis=1)[:, None] + sigma * rng.standard_normal((n, 3))
    return X1, x1, x2, rng.uniform(300, 900, (n, 2)), rng.uniform(1, 4, n)


def odom(rng):
    """an odometry increment with a baseline of at least 0.3 m"""
    while True:
        t = rng.uniform(-0.4, 0.4, 3)
        if np.linalg.norm(t) >= 0.3:
            return np.concatenate([t, rng.uniform(-0.1, 0.1, 3)])


def sparse_errors(res, jac, ref_res, ref_jac, p2, size, what=""):
    """the 8(c) figures on the un-weighted residuals (residual * size = projection - observation), in units of 1e-10"""
    from tests.parity import block_parity_errors

    res, ref_res = np.asarray(res).reshape(-1, 2), np.asarray(ref_res).reshape(-1, 2)
    failed = ref_res[:, 0] == BIG
    assert np.array_equal(res == BIG, ref_res == BIG), "failed-projection pattern differs " + what
    s = np.where(failed, 1.0, np.asarray(size))[:, None]
    J = None if jac is None else [np.asarray(jac).reshape(-1, 6)]
    Jr = None if ref_jac is None else [np.asarray(ref_jac).reshape(-1, 6)]
    return block_parity_errors((res * s).ravel(), J, (ref_res * s).ravel(), Jr, np.asarray(p2).ravel())


def assert_sparse_parity(res, jac, ref_res, ref_jac, p2, size, what="", sens=None):
    """every figure <= 1e-10 -- or, for deliberately ill-conditioned geometry (sens = the same figures between the CPU
    checker at the inputs and at inputs moved by half an ulp, see hard_geometry_sensitivity), <= 1e-10 + 30 x that"""
    errs = sparse_errors(res, jac, ref_res, ref_jac, p2, size, what)
    bad = {k: v * TOL for k, v in errs.items() if not v <= 1.0 + (30.0 * sens.get(k, 0.0) if sens else 0.0)}
    assert not bad, "parity %s: %s (allowed: 1e-10%s)" % (what, bad, " + 30 x %s" % {k: v * TOL for k, v in sens.items()} if sens else "")
    return errs


def hard_geometry_sensitivity(rng, model, xb, x1, x2, p2, size, xo):
    """The triangulated depth is a ratio of two differences of near-equal products (triangulator.cpp:169-175): its condition
    number is ~ depth / (baseline x sin(angle to the epipole)), unbounded near the epipole and for far points.  There two
    correct double evaluations differ by (condition number) x 1e-16.  This measures it: the checker's own change when every
    real input moves by half an ulp (three random sign patterns, worst figure of each kind)."""
    r0, J0 = vgo.sparse_reproject(MODELS[model], INTR[model], xb, x1, x2, p2, size, xo)
    worst = {}
    for _ in range(3):
        e = lambda a: np.asarray(a, float) * (1 + 1.1e-16 * rng.choice([-1.0, 1.0], np.shape(a)))
        r1, J1 = vgo.sparse_reproject(MODELS[model], INTR[model], e(xb), e(x1), e(x2), p2, size, e(xo))
        if not np.array_equal(r1 == BIG, r0 == BIG):
            continue
        for k, v in sparse_errors(r1, J1, r0, J0, p2, size).items():
            worst[k] = max(worst.get(k, 0.0), v)
    return worst


@pytest.mark.parametrize("model", ["eucm", "ucm", "mei"])
def test_mono_reproject_block_equals_the_oracle(loc, model):
    rng = np.random.default_rng(11)
    for trial in range(6):
        xo = odom(rng)
        X1, x1, _, p2, _ = scene(rng, 5, XB, xo)
        lengths = np.linalg.norm(X1, axis=1) * (1 + 0.02 * rng.standard_normal(5))
        if trial == 3:
            lengths[1] = -lengths[1]          # behind the camera: EUCM reports failure, UCM / Mei project anyway
        cost = loc.MonoReprojectCost(model, INTR[model], x1, p2, XB)
        assert cost.num_residuals() == 10 and cost.parameter_block_sizes() == [6, 5]
        r, J = cost.Evaluate([xo, lengths])
        rr, j0, j1 = vgo.mono_reproject(MODELS[model], INTR[model], XB, x1, p2, xo, lengths)
        assert_block_parity(r, J, rr, [j0, j1], p2, "%s mono trial %d" % (model, trial))
        if trial == 3 and model == "eucm":
            assert r[2] == BIG and r[3] == BIG and not J[0][2:4].any() and not J[1][2:4].any()
        # NULL Jacobian blocks / cost only
        r2, J2 = cost.Evaluate([xo, lengths], jac_mask=[False, True])
        assert np.array_equal(r2, r) and J2[0] is None and np.array_equal(J2[1], J[1])
        r3, J3 = cost.Evaluate([xo, lengths], want_jacobians=False)
        assert np.array_equal(r3, r) and J3 is None
        cost.close()


@pytest.mark.parametrize("model", ["eucm", "ucm", "mei"])
def test_sparse_reproject_block_equals_the_oracle(loc, model):
    rng = np.random.default_rng(12)
    for n, far in ((1, 0), (7, 0), (64, 0), (65, 3), (300, 10)):
        xo = odom(rng)
        _, x1, x2, p2, size = scene(rng, n, XB, xo, far=far)
        if n >= 64:                            # pairs that triangulate behind the cameras: EUCM fails them
            x1[5], x2[5] = -x1[5], -x2[5]
        cost = loc.SparseReprojectCost(model, INTR[model], x1, x2, p2, size, XB)
        assert cost.num_residuals() == 2 * n and cost.parameter_block_sizes() == [6]
        r, J = cost.Evaluate([xo])
        rr, Jr = vgo.sparse_reproject(MODELS[model], INTR[model], XB, x1, x2, p2, size, xo)
        sens = hard_geometry_sensitivity(rng, model, XB, x1, x2, p2, size, xo)
        if not far:
            assert max(sens.values()) < 0.5, sens        # a well-conditioned scene: the bar is 1e-10 (+ at most 15 %)
        assert_sparse_parity(r, J[0], rr, Jr, p2, size, "%s sparse n=%d" % (model, n), sens=sens)
        if n >= 64 and model == "eucm":
            assert r[10] == BIG and r[11] == BIG and not J[0][10:12].any()
        r0, J0 = cost.Evaluate([xo], want_jacobians=False)
        assert np.array_equal(r0, r) and J0 is None
        cost.close()


def test_the_regularised_triangulation_is_exercised(loc):
    """far points: delta <= eps * delta1, lambda = 2 / eps - delta / (delta1 eps^2) (triangulator.cpp:114-128) and the
    Jacobian of :209-217 -- the set's rows still equal the oracle's, and the oracle takes that branch for them"""
    rng = np.random.default_rng(13)
    xo = odom(rng)
    n, far = 40, 12
    _, x1, x2, p2, size = scene(rng, n, XB, xo, far=far)
    xi12 = np.linalg.inv(T(XB)) @ T(xo) @ T(XB)
    R, t = xi12[:3, :3], xi12[:3, 3]
    n_reg = 0
    for i in range(n):
        q = R @ x2[i]
        r = x1[i] + q
        delta = (t @ x1[i]) * (r @ q) - (t @ q) * (r @ x1[i])
        delta1 = (t @ t) * (r @ q) - (t @ r) * (t @ q)
        n_reg += not delta > 1e-3 * delta1
    assert n_reg >= far // 2
    cost = loc.SparseReprojectCost("ucm", INTR["ucm"], x1, x2, p2, size, XB)
    r, J = cost.Evaluate([xo])
    rr, Jr = vgo.sparse_reproject(1, INTR["ucm"], XB, x1, x2, p2, size, xo)
    sens = hard_geometry_sensitivity(rng, "ucm", XB, x1, x2, p2, size, xo)
    errs = assert_sparse_parity(r, J[0], rr, Jr, p2, size, "regularised", sens=sens)
    print("far points: HIP vs checker %s; checker vs itself half an ulp away %s" %
          ({k: "%.1e" % (v * TOL) for k, v in errs.items()}, {k: "%.1e" % (v * TOL) for k, v in sens.items()}))
    cost.close()


@pytest.mark.parametrize("model", ["eucm", "mei"])
def test_points_near_the_epipole(loc, model):
    """motion along the viewing direction: the parallax of points near the epipole vanishes and the depth is arbitrarily
    ill-conditioned -- the rows must still agree with the checker as well as the checker agrees with itself half an ulp away"""
    rng = np.random.default_rng(18)
    for trial in range(4):
        T12 = np.array([0.02, -0.01, 0.45, 0.01, -0.02, 0.03])          # camera 2 seen from camera 1: forward motion
        xo_m = T(XB) @ T(T12) @ np.linalg.inv(T(XB))                     # the odometry increment that produces it
        from scipy.spatial.transform import Rotation as Rot

        xo = np.concatenate([xo_m[:3, 3], Rot.from_matrix(xo_m[:3, :3]).as_rotvec()])
        _, x1, x2, p2, size = scene(rng, 60, XB, xo)
        x1[:10] = np.array([0.0, 0.0, 1.0]) + 0.02 * rng.standard_normal((10, 3))   # ten directions next to the epipole
        x1[:10] /= np.linalg.norm(x1[:10], axis=1)[:, None]
        X1 = x1[:10] * rng.uniform(2, 6, (10, 1))
        Tm = np.linalg.inv(T(XB)) @ T(xo) @ T(XB)
        X2 = (np.linalg.inv(Tm) @ np.c_[X1, np.ones(10)].T).T[:, :3]
        x2[:10] = X2 / np.linalg.norm(X2, axis=1)[:, None]
        cost = loc.SparseReprojectCost(model, INTR[model], x1, x2, p2, size, XB)
        r, J = cost.Evaluate([xo])
        rr, Jr = vgo.sparse_reproject(MODELS[model], INTR[model], XB, x1, x2, p2, size, xo)
        sens = hard_geometry_sensitivity(rng, model, XB, x1, x2, p2, size, xo)
        assert_sparse_parity(r, J[0], rr, Jr, p2, size, "epipole trial %d" % trial, sens=sens)
        cost.close()


@pytest.mark.parametrize("model", ["eucm", "mei"])
def test_batched_sets_equal_the_oracle_block_by_block(loc, model):
    """many blocks per launch: 257 five-point blocks; 120 ragged blocks (0 ... 90 points, empty ones included) -- every
    block with its own odometry parameter -- against one oracle call per block, and bit-identical to the per-block entry"""
    import torch

    rng = np.random.default_rng(14)
    nb = 257
    blocks, xos, lens = [], [], []
    for b in range(nb):
        xo = odom(rng)
        X1, x1, _, p2, _ = scene(rng, 5, XB, xo)
        blocks.append((x1, p2))
        xos.append(xo)
        lens.append(np.linalg.norm(X1, axis=1) * (1 + 0.02 * rng.standard_normal(5)))
    st = loc.ReprojectSet(model, INTR[model], XB, blocks, sparse=False)
    xo_t, ln_t = torch.tensor(np.array(xos), device="cuda"), torch.tensor(np.array(lens), device="cuda")
    res, j0, j1 = st.evaluate(xo_t, ln_t)
    st.synchronize()
    res, j0, j1 = res.cpu().numpy(), j0.cpu().numpy(), j1.cpu().numpy()
    for b in range(nb):
        rr, r0, r1 = vgo.mono_reproject(MODELS[model], INTR[model], XB, blocks[b][0], blocks[b][1], xos[b], lens[b])
        assert_block_parity(res[b], [j0[b], j1[b]], rr, [r0, r1], blocks[b][1], "mono block %d" % b)
    for b in (0, 100, 256):
        r, J = st.evaluate_block(b, [xos[b], lens[b]])
        assert np.array_equal(r, res[b]) and np.array_equal(J[0], j0[b]) and np.array_equal(J[1], j1[b])
    st.close()

    nb = 120
    counts = rng.integers(0, 91, nb)
    counts[[3, 50, nb - 1]] = 0
    blocks, xos = [], []
    for b in range(nb):
        xo = odom(rng)
        _, x1, x2, p2, size = scene(rng, int(counts[b]), XB, xo)
        blocks.append((x1, x2, p2, size))
        xos.append(xo)
    st = loc.ReprojectSet(model, INTR[model], XB, blocks, sparse=True)
    assert st.n_points == counts.sum() and st.n_blocks == nb
    res, jac = st.evaluate(torch.tensor(np.array(xos), device="cuda"))
    st.synchronize()
    res, jac = res.cpu().numpy(), jac.cpu().numpy()
    n_strict = 0
    for b in range(nb):
        lo, hi = st.offsets[b], st.offsets[b + 1]
        if hi == lo:
            continue
        rr, Jr = vgo.sparse_reproject(MODELS[model], INTR[model], XB, *blocks[b], xos[b])
        sens = hard_geometry_sensitivity(rng, model, XB, *blocks[b], xos[b])
        n_strict += max(sens.values()) < 0.05
        assert_sparse_parity(res[lo:hi], jac[lo:hi], rr, Jr, blocks[b][2], blocks[b][3], "sparse block %d" % b, sens=sens)
    assert n_strict > nb // 2      # most blocks are well conditioned: their bar is 1e-10 (+ at most 1.5 %)
    for b in (0, 3, 77):
        r, J = st.evaluate_block(b, [xos[b]])
        lo, hi = st.offsets[b], st.offsets[b + 1]
        assert np.array_equal(r, res[lo:hi].ravel()) and np.array_equal(J[0], jac[lo:hi].reshape(-1, 6))
    res2, none = st.evaluate(np.array(xos), want_jac=False)
    st.synchronize()
    assert none is None and np.array_equal(res2.cpu().numpy(), res)
    st.close()


@pytest.mark.parametrize("model", ["eucm", "ucm", "mei"])
@pytest.mark.parametrize("two", [False, True])
def test_camera_jacobian_equals_the_oracle(loc, model, two):
    rng = np.random.default_rng(15)
    T12 = np.array([0.3, -0.2, 0.1, 0.4, -0.3, 0.2])
    T23 = np.array([-0.1, 0.25, 0.05, -0.2, 0.1, 0.5]) if two else None
    n = 1000
    X = np.column_stack([rng.uniform(-1, 1, n), rng.uniform(-0.7, 0.7, n), rng.uniform(1.5, 5, n)])
    X[17] = [0.1, 0.2, -3.0]                  # EUCM cannot project it: zero rows (jacobian.h:78-83)
    grad = rng.standard_normal((n, 2))
    dp, df = loc.camera_jacobian(model, INTR[model], T12, T23, X, grad)
    dp, df = dp.cpu().numpy(), df.cpu().numpy()
    du, dv, dfr = vgo.camera_jacobian(MODELS[model], INTR[model], T12, T23, X, grad)
    ref = np.stack([du, dv], axis=1)
    for got, want in ((dp, ref), (df, dfr)):
        assert np.linalg.norm(got - want) <= TOL * np.linalg.norm(want)
        g2, w2 = got.reshape(n, -1), want.reshape(n, -1)
        floor = 1e-3 * np.max(np.abs(w2), axis=1, keepdims=True)
        assert np.max(np.abs(g2 - w2) / np.maximum(np.abs(w2), np.maximum(floor, 1e-300))) <= TOL
    if model == "eucm":
        assert not dp[17].any() and not df[17].any() and not ref[17].any()
    only_dp, none = loc.camera_jacobian(model, INTR[model], T12, T23, X)
    assert none is None and np.array_equal(only_dp.cpu().numpy(), dp)


def test_a_ransac_sized_and_a_large_set(loc):
    """SparseOdometry::ransacNPoints (sparse_odom.cpp:511-606): 200 hypotheses of a few points each -- one launch pair;
    and 2 000 blocks x 500 points = 1 M features: the batched pass equals the per-block entry bit for bit on sampled
    blocks, residuals are finite, every (u-row) Jacobian row is the oracle's on a sample"""
    import time

    import torch

    rng = np.random.default_rng(16)
    xo_true = odom(rng)
    _, x1, x2, p2, size = scene(rng, 400, XB, xo_true)
    blocks, xos = [], []
    for h in range(200):
        idx = rng.permutation(400)[:8]
        blocks.append((x1[idx], x2[idx], p2[idx], size[idx]))
        xos.append(xo_true + 0.01 * rng.standard_normal(6))
    st = loc.ReprojectSet("eucm", INTR["eucm"], XB, blocks, sparse=True)
    xo_t = torch.tensor(np.array(xos), device="cuda")
    st.evaluate(xo_t)
    st.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        res, jac = st.evaluate(xo_t)
    st.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print("200 hypotheses x 8 points: %.1f us per evaluation of all blocks" % (dt * 1e6))
    for h in (0, 99, 199):
        rr, Jr = vgo.sparse_reproject(0, INTR["eucm"], XB, *blocks[h], xos[h])
        assert_sparse_parity(res[8 * h:8 * h + 8].cpu().numpy(), jac[8 * h:8 * h + 8].cpu().numpy(), rr, Jr, blocks[h][2], blocks[h][3],
                             sens=hard_geometry_sensitivity(rng, "eucm", XB, *blocks[h], xos[h]))
    st.close()

    nb, per = 2000, 500
    _, x1, x2, p2, size = scene(rng, per, XB, xo_true)
    blocks = [(x1, x2, p2, size)] * nb
    xos = xo_true + 0.01 * rng.standard_normal((nb, 6))
    st = loc.ReprojectSet("mei", INTR["mei"], XB, blocks, sparse=True)
    xo_t = torch.tensor(xos, device="cuda")
    st.evaluate(xo_t)
    st.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        res, jac = st.evaluate(xo_t)
    st.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("1 M features (2 000 blocks x 500): %.1f us per evaluation, %.2e features/s, %.0f GB/s of rows" %
          (dt * 1e6, nb * per / dt, nb * per * (16 + 96 + 72) / dt / 1e9))
    res, jac = res.cpu().numpy(), jac.cpu().numpy()
    assert np.isfinite(res).all() and np.isfinite(jac).all()
    for b in (0, 777, 1999):
        r, J = st.evaluate_block(b, [xos[b]])
        assert np.array_equal(r, res[per * b:per * (b + 1)].ravel()) and np.array_equal(J[0], jac[per * b:per * (b + 1)].reshape(-1, 6))
        rr, Jr = vgo.sparse_reproject(2, INTR["mei"], XB, x1, x2, p2, size, xos[b])
        assert_sparse_parity(r, J[0], rr, Jr, p2, size, "large block %d" % b, sens=hard_geometry_sensitivity(rng, "mei", XB, x1, x2, p2, size, xos[b]))
    st.close()


def test_many_tiny_blocks_take_the_frame_launch(loc):
    """a workgroup's 256 points spread over more blocks than the in-kernel frame table holds (300 one- and two-point blocks,
    some empty): the two-launch route on a small grid; equal to the per-block entry (one block per launch: frames computed
    in the kernel) bit for bit, and to the oracle"""
    import torch

    rng = np.random.default_rng(18)
    nb = 300
    counts = rng.integers(0, 3, nb)
    blocks, xos = [], []
    for b in range(nb):
        xo = odom(rng)
        _, x1, x2, p2, size = scene(rng, int(counts[b]), XB, xo)
        blocks.append((x1, x2, p2, size))
        xos.append(xo)
    st = loc.ReprojectSet("ucm", INTR["ucm"], XB, blocks, sparse=True)
    res, jac = st.evaluate(torch.tensor(np.array(xos), device="cuda"))
    st.synchronize()
    res, jac = res.cpu().numpy(), jac.cpu().numpy()
    for b in range(nb):
        lo, hi = st.offsets[b], st.offsets[b + 1]
        if hi == lo:
            continue
        r, J = st.evaluate_block(b, [xos[b]])
        assert np.array_equal(r, res[lo:hi].ravel()) and np.array_equal(J[0], jac[lo:hi].reshape(-1, 6)), b
        if b % 10 == 0:
            rr, Jr = vgo.sparse_reproject(MODELS["ucm"], INTR["ucm"], XB, *blocks[b], xos[b])
            assert_sparse_parity(res[lo:hi], jac[lo:hi], rr, Jr, blocks[b][2], blocks[b][3], "tiny block %d" % b,
                                 sens=hard_geometry_sensitivity(rng, "ucm", XB, *blocks[b], xos[b]))
    st.close()


def test_argument_errors(loc):
    from visgeom_amd import capi

    rng = np.random.default_rng(17)
    _, x1, x2, p2, size = scene(rng, 5, XB, np.zeros(6))
    with pytest.raises(ValueError):
        loc.MonoReprojectCost("eucm", INTR["eucm"], x1[:4], p2[:4], XB)        # the reference asserts five points
    with pytest.raises(ValueError):
        loc.SparseReprojectCost("eucm", INTR["eucm"], x1, x2[:4], p2, size, XB)
    with pytest.raises(ValueError):
        loc.MonoReprojectCost("eucm", INTR["ucm"], x1, p2, XB)                  # wrong number of intrinsics
    c = loc.MonoReprojectCost("ucm", INTR["ucm"], x1, p2, XB)
    with pytest.raises(ValueError):
        c.Evaluate([np.zeros(6)])
    L = capi.load()
    assert L.vg_mono_reproject_block_evaluate(c._set._h, 1, None, None, None) == capi.ERR_INVALID_ARGUMENT
    assert L.vg_sparse_reproject_evaluate(c._set._h, None, None, None) == capi.ERR_INVALID_ARGUMENT   # a mono set
    assert L.vg_reproject_num_blocks(c._set._h) == 1 and L.vg_reproject_num_points(c._set._h) == 5
    assert L.vg_reproject_block_offset(c._set._h, 1) == 5 and L.vg_reproject_block_offset(c._set._h, 2) == -1
    c.close()
    empty = loc.ReprojectSet("eucm", INTR["eucm"], XB, [], sparse=True)
    assert empty.n_blocks == 0 and empty.n_points == 0
    res, jac = empty.evaluate(np.zeros((0, 6)))
    assert tuple(res.shape) == (0, 2)
    empty.close()
