"""The geometric pose initialisation of the front end, block by block, WITHOUT a GPU: the product's
host code -- the same functions vg_calibration_add_file runs, exported as vg_reconstruct_point / vg_initial_grid_pose /
vg_init_transform -- against the oracle's restatement of ICamera::reconstructPoint (eucm.h:85-106, ucm.h:81-103,
mei.h:90-112), estimateInitialGrid's 4-corner construction (src/calibration/unified_calibration.cpp:1066-1135) and
getInitTransform (:311-348).  (VERDICT r2 missing 5: these existed in the product only.)"""
import ctypes

import numpy as np
import pytest

from oracle import vgo
from visgeom_amd import capi, synthetic as S

MODELS = {"eucm": 0, "ucm": 1, "mei": 2}
_dp = ctypes.POINTER(ctypes.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


@pytest.fixture(scope="module")
def lib():
    return capi.load()


def product_reconstruct(lib, model, intr, uv):
    intr, uv, X = np.ascontiguousarray(intr, float), np.ascontiguousarray(uv, float), np.full(3, np.nan)
    return lib.vg_reconstruct_point(MODELS[model], _p(intr), _p(uv), _p(X)) == 0, X


def product_grid_pose(lib, model, intr, board4, corners4):
    intr = np.ascontiguousarray(intr, float)
    b, c, xi = np.ascontiguousarray(board4, float).reshape(12), np.ascontiguousarray(corners4, float).reshape(8), np.full(6, np.nan)
    return lib.vg_initial_grid_pose(MODELS[model], _p(intr), _p(b), _p(c), _p(xi)) == 0, xi


def product_init_transform(lib, status, init_index, chain, xi):
    chain, xi, out = np.ascontiguousarray(chain, float).reshape(-1, 6), np.ascontiguousarray(xi, float), np.empty(6)
    st = (ctypes.c_int * max(len(status), 1))(*status)
    assert lib.vg_init_transform(chain.shape[0], st, init_index, _p(chain), _p(xi), _p(out)) == 0
    return out


@pytest.mark.parametrize("model", ["eucm", "ucm", "mei"])
def test_reconstruct_point_equals_the_oracle_and_inverts_the_projection(lib, model):
    rng = np.random.default_rng(3)
    intr = S.GT[model]
    for _ in range(200):
        uv = np.array([rng.uniform(0, S.IMAGE_W), rng.uniform(0, S.IMAGE_H)])
        ok, X = product_reconstruct(lib, model, intr, uv)
        ok_o, X_o = vgo.reconstruct_point(MODELS[model], intr, uv)
        assert ok == ok_o
        if ok and not np.isfinite(X_o).all():   # UCM / Mei outside the image circle: sqrt of a negative number, in both
            assert np.array_equal(np.isnan(X), np.isnan(X_o))
            continue
        if ok:
            assert np.max(np.abs(X - X_o)) <= 1e-14 * max(1.0, np.max(np.abs(X_o)))
            if model != "mei":        # Mei's reconstruction ignores its distortion terms (mei.h:90-112)
                okp, back = vgo.project_point(MODELS[model], intr, X)
                assert okp and np.max(np.abs(back - uv)) < 1e-9
    # EUCM: beyond the model's image circle det < 0 -> false (eucm.h:100)
    far = np.array([S.GT["eucm"][4] + 40 * S.GT["eucm"][2], S.GT["eucm"][5]])
    assert not product_reconstruct(lib, "eucm", S.GT["eucm"], far)[0] and not vgo.reconstruct_point(0, S.GT["eucm"], far)[0]


@pytest.mark.parametrize("model", ["eucm", "ucm", "mei"])
def test_four_corner_pose_equals_the_oracle_and_is_near_the_generating_pose(lib, model):
    """every image of a synthetic set: the product's construction against the oracle's (the oracle forms (I - ex ex^T) ey as
    the reference does, the product ey - ex (ex . ey): rounding-level differences only), and both land near the pose the
    image was generated from -- at the generating intrinsics, noise-free corners"""
    d = S.make_mono(model, 60, 2, sigma=0.0)
    N = d["board"].shape[0]
    idx = [0, S.BOARD_COLS - 1, N - S.BOARD_COLS, N - 1]           # UL, UR, BL, BR of the 12 x 8 board
    worst = 0.0
    for i in range(60):
        ok, xi = product_grid_pose(lib, model, d["gt_intrinsics"], d["board"][idx], d["corners"][i][idx])
        ok_o, xi_o = vgo.initial_grid_pose(MODELS[model], d["gt_intrinsics"], d["board"][idx], d["corners"][i][idx])
        assert ok and ok_o
        assert np.max(np.abs(xi - xi_o)) <= 1e-12, (i, xi, xi_o)
        # a starting value for the refinement, not an estimate: unit rays scaled by the smaller of two edge ratios
        R = vgo.rotation_matrix(d["gt_poses"][i][3:])
        ul_cam = R @ d["board"][idx[0]] + d["gt_poses"][i][:3]
        # (the translation is the UL corner's viewing ray -- exact -- times a range that is off by the board's obliquity)
        cosang = xi[:3] @ ul_cam / (np.linalg.norm(xi[:3]) * np.linalg.norm(ul_cam))
        assert cosang > 1 - (1e-9 if model != "mei" else 1e-3) and 0.6 < np.linalg.norm(xi[:3]) / np.linalg.norm(ul_cam) < 1.4
        worst = max(worst, np.linalg.norm(vgo.rotation_matrix(xi[3:]) - R))
    assert worst < 0.8
    bad = d["corners"][0][idx].copy()
    if model == "eucm":
        bad[2] = [1e6, 1e6]
        assert not product_grid_pose(lib, model, d["gt_intrinsics"], d["board"][idx], bad)[0]
        assert not vgo.initial_grid_pose(0, d["gt_intrinsics"], d["board"][idx], bad)[0]


def test_rotation_vector_of_a_matrix(lib):
    rng = np.random.default_rng(4)
    for _ in range(50):
        r = rng.standard_normal(3)
        r *= rng.uniform(0.01, 2.5) / np.linalg.norm(r)
        R = vgo.rotation_matrix(r)
        assert np.max(np.abs(vgo.rotation_vector(R) - r)) < 1e-12
        out = np.empty(6)
        vals = np.ascontiguousarray(np.concatenate([np.c_[R, [0.1, 0.2, 0.3]].ravel()]))      # 12 values: row-major [R | t]
        assert lib.vg_transform_from_values(12, _p(vals), _p(out)) == 0
        assert np.max(np.abs(out[3:] - vgo.rotation_vector(R))) <= 1e-15 and np.allclose(out[:3], [0.1, 0.2, 0.3])


def test_init_transform_equals_the_oracle_and_recovers_the_member(lib):
    """chains of 1..5 members with every status pattern tried at random: the member being initialised, recovered from the
    camera-frame pose that the FULL chain produces, must be the member's own value -- and the product's peeling must equal
    the oracle's"""
    rng = np.random.default_rng(5)
    for trial in range(80):
        L = int(rng.integers(1, 6))
        status = [int(s) for s in rng.integers(0, 2, L)]
        chain = np.concatenate([rng.uniform(-0.5, 0.5, (L, 3)), rng.uniform(-0.6, 0.6, (L, 3))], axis=1)
        init = int(rng.integers(0, L))
        # the camera-frame pose of the board through the whole chain (calib_cost_functions.cpp:32-46)
        acc = np.zeros(6)
        for l in range(L):
            acc = vgo.compose(acc, chain[l], inverse=bool(status[l]))
        got = product_init_transform(lib, status, init, chain, acc)
        ref = vgo.init_transform(status, init, chain, acc)
        assert np.max(np.abs(got - ref)) <= 1e-13, (status, init)
        R_got, R_own = vgo.rotation_matrix(got[3:]), vgo.rotation_matrix(chain[init][3:])
        assert np.max(np.abs(R_got - R_own)) < 1e-10 and np.max(np.abs(got[:3] - chain[init][:3])) < 1e-10, (status, init)
    # a name that is not in the chain (init_index == chain length): both loops peel every member, as the reference does
    chain = np.array([[0.1, 0.2, 0.3, 0.01, 0.02, 0.03]])
    xi = np.array([0.5, 0.1, 1.0, 0.1, 0.2, 0.3])
    assert np.max(np.abs(product_init_transform(lib, [0], 1, chain, xi) - vgo.init_transform([0], 1, chain, xi))) <= 1e-13


def test_init_transform_with_a_member_that_occurs_twice(lib):
    """unified_calibration.cpp:311-348 compares NAMES: when the member being initialised occurs twice in a chain the forward
    loop stops at its first occurrence and the backward loop at its LAST one; the members in between are visited by
    neither loop (the product kept only the first occurrence)"""
    rng = np.random.default_rng(11)
    for trial in range(40):
        L = int(rng.integers(2, 6))
        status = [int(s) for s in rng.integers(0, 2, L)]
        chain = np.concatenate([rng.uniform(-0.5, 0.5, (L, 3)), rng.uniform(-0.6, 0.6, (L, 3))], axis=1)
        first = int(rng.integers(0, L - 1))
        last = int(rng.integers(first + 1, L))
        xi = np.concatenate([rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.6, 0.6, 3)])
        st = (ctypes.c_int * L)(*status)
        got = np.empty(6)
        assert lib.vg_init_transform_range(L, st, first, last, _p(np.ascontiguousarray(chain)), _p(xi), _p(got)) == 0
        ref = vgo.init_transform_range(status, first, last, chain, xi)
        assert np.max(np.abs(got - ref)) <= 1e-13, (status, first, last)
        # independent restatement with the oracle's compose primitives: members before `first` peeled from the left, members
        # after `last` from the right, inversion when the LAST occurrence is used INVERSE
        acc = xi.copy()
        inv = lambda a: vgo.compose(np.zeros(6), a, inverse=True)
        for i in range(first):
            acc = vgo.compose(inv(chain[i]), acc) if status[i] == 0 else vgo.compose(chain[i], acc)
        for i in range(L - 1, last, -1):
            acc = vgo.compose(acc, chain[i], inverse=(status[i] == 0))
        if status[last] == 1:
            acc = inv(acc)
        Ra, Rg = vgo.rotation_matrix(acc[3:]), vgo.rotation_matrix(got[3:])
        assert np.max(np.abs(Ra - Rg)) < 1e-12 and np.max(np.abs(acc[:3] - got[:3])) < 1e-12, (status, first, last)
