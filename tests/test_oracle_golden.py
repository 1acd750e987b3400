"""Pin the CPU oracle against the reference's own known answers.

The vectors in tests/golden/survey_appendix_c.json were produced at survey time by the
reference's code (GenericProjectionJac::Evaluate, src/calibration/calib_cost_functions.cpp:28-117);
they are the only outputs of the reference that exist for this path (the reference has no tests,
SURVEY.md section 4, and cannot be built in this image).
"""
import json
import os

import numpy as np
import pytest

from oracle import vgo

TOL = 1e-13  # the oracle reproduces the vectors bit for bit; 1e-13 leaves room for libm changes


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "survey_appendix_c.json")) as f:
        return json.load(f)


def board(b):
    # ordering of unified_calibration.cpp:286-292: k = i*cols + j -> (size*j, size*i, 0)
    return np.array([[b["size"] * j, b["size"] * i, 0.0] for i in range(b["rows"]) for j in range(b["cols"])])


def relerr(a, ref):
    a, ref = np.asarray(a, float), np.asarray(ref, float)
    return np.max(np.abs(a - ref) / np.maximum(np.abs(ref), 1e-3 * np.max(np.abs(ref)) + 1e-300))


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_c1_single_point_inverse_direct_chain(golden, idx):
    c1 = golden["c1"]
    case = c1["cases"][idx]
    model = vgo.MODELS[case["model"]]
    res, J = vgo.eval_block(model, c1["status"], c1["grid"], c1["obs"],
                            [case["intrinsics"], c1["xi12"], c1["xiB"]])
    assert relerr(res, case["residual"]) < TOL
    assert np.max(np.abs(J[0][0] - np.array(case["du_dintr"]))) < TOL * 10
    assert relerr(J[1][0], case["du_dxi12"]) < TOL
    assert relerr(J[2][1], case["dv_dxiB"]) < TOL


def test_c2_full_block(golden):
    c2 = golden["c2"]
    grid = board(c2["board"])
    model = vgo.MODELS[c2["model"]]
    proj, _ = vgo.eval_block(model, c2["status"], grid, np.zeros((96, 2)), [c2["intrinsics"], c2["xi"]],
                             want_jac=False)
    obs = proj.reshape(-1, 2) + np.array(c2["obs_offset"])
    res, J = vgo.eval_block(model, c2["status"], grid, obs, [c2["intrinsics"], c2["xi"]])
    assert res.size == c2["num_residuals"]
    assert [J[0].shape[1], J[1].shape[1]] == c2["block_sizes"]
    assert np.max(np.abs(res.reshape(-1, 2) - np.array(c2["every_residual_pair"]))) < 1e-10
    assert np.max(np.abs(J[0][0] - np.array(c2["intr_jac_row0"]))) < 1e-12
    assert relerr(J[1][191], c2["pose_jac_row191"]) < TOL


def test_c3_zero_rotation_small_angle_branches(golden):
    c3 = golden["c3"]
    z = c3["zero_rotation"]
    grid = board(c3["board"])
    model = vgo.MODELS[c3["model"]]
    proj, _ = vgo.eval_block(model, [0], grid, np.zeros((96, 2)), [c3["intrinsics"], z["obs_from_xi"]],
                             want_jac=False)
    res, J = vgo.eval_block(model, [0], grid, proj.reshape(-1, 2), [c3["intrinsics"], z["xi"]])
    assert relerr(res[2:4], z["residual_2_3"]) < TOL
    assert np.max(np.abs(J[1][2] - np.array(z["pose_jac_row2"]))) < 1e-11


def test_c3_board_behind_camera_inband_failure(golden):
    """Failed EUCM projection: residual pair = 1e15, Jacobian rows = 0 (calib_cost_functions.cpp:66-70,
    eucm.h:141-150,198-206).  Appendix C.3 says "all" corners fail for pose [0,0,-1,0,0,0]; working the
    reference's two tests (eucm.h:40-48) by hand shows that corners (j=11,i=6) and (j=11,i=7) pass them
    (eta = 0.481 / 0.501 > 1e-3 and z/eta = -2.08 / -2.00 >= C = (alpha-1)/(2 alpha-1) = -2.11), so the
    statement holds for 94 of the 96 corners; the two exceptions are asserted explicitly."""
    c3 = golden["c3"]
    grid = board(c3["board"])
    model = vgo.MODELS[c3["model"]]
    res, J = vgo.eval_block(model, [0], grid, np.zeros((96, 2)), [c3["intrinsics"], c3["behind_camera"]["xi"]])
    failed = (res.reshape(-1, 2) == 1e15).all(axis=1)
    passing = sorted(np.nonzero(~failed)[0].tolist())
    assert passing == [6 * 12 + 11, 7 * 12 + 11]
    rows = np.repeat(failed, 2)
    assert np.all(res[rows] == c3["behind_camera"]["all_residuals"])
    assert np.all(J[0][rows] == 0) and np.all(J[1][rows] == 0)
    assert np.all(np.isfinite(J[0])) and np.all(np.isfinite(J[1]))


def test_null_jacobian_blocks_are_skipped(golden):
    """jacobian == NULL (cost only) and jacobian[b] == NULL (constant block): :73,93,105."""
    c1 = golden["c1"]
    case = c1["cases"][0]
    model = vgo.MODELS[case["model"]]
    params = [case["intrinsics"], c1["xi12"], c1["xiB"]]
    res0, J0 = vgo.eval_block(model, c1["status"], c1["grid"], c1["obs"], params, want_jac=False)
    assert J0 is None
    res1, J1 = vgo.eval_block(model, c1["status"], c1["grid"], c1["obs"], params, jac_mask=[False, True, False])
    assert np.array_equal(res0, res1)
    assert J1[0] is None and J1[2] is None
    assert relerr(J1[1][0], case["du_dxi12"]) < TOL
