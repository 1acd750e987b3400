"""vg_refine_poses: the per-image pose refinement of estimateInitialGrid
(src/calibration/unified_calibration.cpp:1137-1155) as n independent problems in one launch.  Every image must end at
ITS OWN least-squares optimum (per-image scipy solve on the oracle; with one residual block per problem SoftLOneLoss is
a monotone function of the plain cost, so the minimiser is the unweighted one), whatever its neighbours do."""
import numpy as np
import pytest

from oracle import vgo

pytestmark = pytest.mark.gpu


def scipy_pose(model, intr, board, corners, x0):
    from scipy.optimize import least_squares

    m = vgo.MODELS[model]

    def fun(x):
        return vgo.eval_block(m, [0], board, corners, [intr, x], want_jac=False)[0]

    def jac(x):
        return vgo.eval_block(m, [0], board, corners, [intr, x])[1][1]

    return least_squares(fun, x0, jac=jac, method="trf", x_scale="jac", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=200)


@pytest.mark.parametrize("model", ["eucm", "ucm", "mei"])
def test_every_image_reaches_its_own_optimum(model):
    from visgeom_amd import synthetic as S
    from visgeom_amd.calibration import refine_poses

    n = 37
    d = S.make_mono(model, n, 3)
    intr = d["gt_intrinsics"] * (1 + 1e-3)       # intrinsics are held constant, slightly off the generating ones
    start = d["gt_poses"] + np.random.default_rng(3).uniform(-0.03, 0.03, (n, 6))
    poses, it, cost, term = refine_poses(model, intr, d["board"], d["corners"], start)
    assert np.all(it >= 2) and np.all(it < 100) and np.all(term <= 2)       # converged, each on its own count
    assert len(set(it.tolist())) > 1, "independent problems do not all stop at the same iteration"
    for b in range(0, n, 4):
        ref = scipy_pose(model, intr, d["board"], d["corners"][b], start[b])
        assert np.max(np.abs(poses[b] - ref.x)) < 1e-6, (b, poses[b], ref.x)
        s = 2 * ref.cost
        assert abs(cost[b] - 0.5 * 2 * 625 * (np.sqrt(1 + s / 625) - 1)) <= 1e-6 * max(cost[b], 1e-12)  # rho(s) / 2, a = 25


def test_a_poisoned_image_does_not_touch_its_neighbours():
    """one image whose start puts the board behind the camera (1e15 residuals, zero Jacobian rows) among good ones:
    the others must end exactly where they end without it (one joint trust region coupled them)"""
    from visgeom_amd import synthetic as S
    from visgeom_amd.calibration import refine_poses

    n = 24
    d = S.make_mono("eucm", n, 2)
    start = d["gt_poses"] + 0.01
    bad = start.copy()
    bad[5] = [0, 0, -1.0, 0, 0, 0]
    bad[17, 3:] += 2.5                      # a far-off start: many iterations or a local minimum, its own business
    good, it_g, cost_g, _ = refine_poses("eucm", d["gt_intrinsics"], d["board"], d["corners"], start)
    mixed, it_m, cost_m, term_m = refine_poses("eucm", d["gt_intrinsics"], d["board"], d["corners"], bad)
    keep = [b for b in range(n) if b not in (5, 17)]
    assert np.array_equal(good[keep], mixed[keep]) and np.array_equal(it_g[keep], it_m[keep])
    assert np.all(np.isfinite(mixed)) and np.all(np.isfinite(cost_m))    # the poisoned image itself: whatever its two projecting corners allow
    assert np.max(np.abs(good[keep] - d["gt_poses"][keep])) < 5e-3


def test_options_and_ragged_sizes():
    from visgeom_amd import synthetic as S
    from visgeom_amd.calibration import refine_poses

    rng = np.random.default_rng(9)
    d = S.make_mono("ucm", 5, 2)
    # 41 of the 96 corners (N not a multiple of 32), 5 images (not a multiple of 8), no loss function, one iteration
    idx = np.sort(rng.choice(96, 41, replace=False))
    start = d["gt_poses"] + 0.02
    p1, it, cost, term = refine_poses("ucm", d["gt_intrinsics"], d["board"][idx], d["corners"][:, idx], start,
                                      max_num_iterations=1, soft_l1_scale=0.0)
    assert np.all(it == 1) and np.all(term == 3)
    pN, itN, costN, _ = refine_poses("ucm", d["gt_intrinsics"], d["board"][idx], d["corners"][:, idx], start, soft_l1_scale=0.0)
    assert np.all(costN < cost) and np.all(itN > 1)
    for b in range(5):
        ref = scipy_pose("ucm", d["gt_intrinsics"], d["board"][idx], d["corners"][b, idx], start[b])
        assert np.max(np.abs(pN[b] - ref.x)) < 1e-6 and abs(costN[b] - ref.cost) <= 1e-6 * ref.cost


@pytest.mark.parametrize("model", ["eucm", "mei"])
def test_resident_route_gives_the_same_bits_as_the_host_pointer_route(model):
    """vg_dataset_refine_poses (the dataset's corners, board and the camera's current intrinsics read in HBM) against
    vg_refine_poses (everything uploaded per call): same kernel, same inputs -> same poses, counts, costs, bit for bit; repeated
    calls (the library's cached scratch) and a second problem of another size in between do not change them."""
    from visgeom_amd import CalibrationProblem, synthetic as S
    from visgeom_amd.calibration import refine_poses

    n = 203   # not a multiple of the 8 images per workgroup
    d = S.make_mono(model, n, 5)
    intr = d["gt_intrinsics"] * (1 - 5e-4)
    start = d["gt_poses"] + np.random.default_rng(11).uniform(-0.02, 0.02, (n, 6))
    ref = refine_poses(model, intr, d["board"], d["corners"], start)
    p = CalibrationProblem(0)
    cam = p.add_camera(model, intr)
    # the dataset's own chain is irrelevant to the refinement (camera-frame poses): a two-member chain here
    g = p.add_transform(True, np.array([0.1, 0, 0, 0, 0.01, 0]))
    seq = p.add_transform(False, start)
    ds = p.add_dataset(cam, [(g, 1), (seq, 0)], d["board"], d["corners"])
    p.finalize()
    for rep in range(3):
        ks = []
        got = p.refine_poses(ds, start, kernel_seconds=ks)
        assert ks[0] > 0
        for a, b in zip(got, ref):
            assert a.tobytes() == b.tobytes()
        if rep == 0:   # another call of another size through the same cached scratch
            small = refine_poses(model, intr, d["board"], d["corners"][:5], start[:5])
            assert small[0].tobytes() == ref[0][:5].tobytes()
    # the intrinsics are the problem's CURRENT ones
    x = p.get_parameters()
    x[p.camera_offset(cam):p.camera_offset(cam) + intr.size] = d["gt_intrinsics"]
    p.set_parameters(x)
    got2 = p.refine_poses(ds, start)
    ref2 = refine_poses(model, d["gt_intrinsics"], d["board"], d["corners"], start)
    assert got2[0].tobytes() == ref2[0].tobytes() and got2[0].tobytes() != ref[0].tobytes()
    p.close()


def test_public_add_dataset_rejects_missing_corners():
    """corners == NULL with images is an error on the public entry (the zero-observation projection dataset of
    writeImageResidual is an internal entry)"""
    import ctypes

    from visgeom_amd import capi, synthetic as S

    L = capi.load()
    d = S.make_mono("eucm", 3, 1)
    h = ctypes.c_void_p()
    capi.check(L.vg_problem_create(ctypes.byref(h), 0, None))
    cam, seq = ctypes.c_int(-1), ctypes.c_int(-1)
    dp = ctypes.POINTER(ctypes.c_double)
    capi.check(L.vg_problem_add_camera(h, 0, d["init_intrinsics"].ctypes.data_as(dp), 0, ctypes.byref(cam)))
    capi.check(L.vg_problem_add_transform(h, 0, 0, 3, np.ascontiguousarray(d["init_poses"]).ctypes.data_as(dp), ctypes.byref(seq)))
    tids, st = (ctypes.c_int * 1)(seq.value), (ctypes.c_int * 1)(0)
    board = np.ascontiguousarray(d["board"])
    rc = L.vg_problem_add_dataset(h, cam.value, 1, tids, st, board.shape[0], board.ctypes.data_as(dp), 3, None, None, None)
    assert rc == capi.ERR_INVALID_ARGUMENT and b"corners" in L.vg_last_error()
    L.vg_problem_destroy(h)


def test_cached_scratch_survives_release_and_growth():
    """the refinement's device / pinned blocks are kept between calls and handed back by vg_release_cached_memory: a call after the
    release, a larger call (the blocks grow) and a smaller one again give the bits of the first call"""
    import visgeom_amd
    from visgeom_amd import synthetic as S
    from visgeom_amd.calibration import refine_poses

    d = S.make_mono("ucm", 300, 4)
    start = d["gt_poses"] + 0.01
    a = refine_poses("ucm", d["gt_intrinsics"], d["board"], d["corners"][:40], start[:40])
    visgeom_amd.release_cached_memory()
    b = refine_poses("ucm", d["gt_intrinsics"], d["board"], d["corners"][:40], start[:40])
    big = refine_poses("ucm", d["gt_intrinsics"], d["board"], d["corners"], start)
    c = refine_poses("ucm", d["gt_intrinsics"], d["board"], d["corners"][:40], start[:40])
    for x, y, z, w in zip(a, b, c, big):
        assert x.tobytes() == y.tobytes() == z.tobytes() == w[:40].tobytes()
