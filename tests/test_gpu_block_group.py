"""Block groups: the per-block drop-in (vg_block_evaluate <-> GenericProjectionJac::Evaluate) amortised over all blocks
of a problem.  The tests play the host: a Ceres-like evaluator that keeps its candidate points in state arrays of a
fixed layout (x / x_plus_delta) and hands every cost function pointers into the array of the current pass."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Host:
    """parameter blocks at fixed offsets of a state array; constant blocks stay in 'user' memory"""

    def __init__(self, K, n_poses, n_glob=0, constant_intrinsics=False):
        self.K, self.n, self.n_glob, self.const_intr = K, n_poses, n_glob, constant_intrinsics
        self.size = (0 if constant_intrinsics else K) + 6 * n_glob + 6 * n_poses
        self.user_intr = np.zeros(K)

    def views(self, state):
        o = 0
        if self.const_intr:
            intr = self.user_intr
        else:
            intr = state[0:self.K]
            o = self.K
        glob = [state[o + 6 * g:o + 6 * g + 6] for g in range(self.n_glob)]
        o += 6 * self.n_glob
        poses = [state[o + 6 * i:o + 6 * i + 6] for i in range(self.n)]
        return intr, glob, poses


def make(vg, model, n, mode, constant_intrinsics=False, seed=3):
    from visgeom_amd import synthetic as S

    d = S.make_mono(model, n, 2)
    K = d["init_intrinsics"].size
    host = Host(K, n, 0, constant_intrinsics)
    group = vg.BlockGroup(0, mode)
    blocks = [vg.GenericProjectionJac(d["corners"][i], d["board"], model, [0], group=group) for i in range(n)]
    plain = [vg.GenericProjectionJac(d["corners"][i], d["board"], model, [0]) for i in range(n)]
    rng = np.random.default_rng(seed)

    def fill(state, scale):
        intr, _, poses = host.views(state)
        intr[:] = d["init_intrinsics"] * (1 + scale * rng.standard_normal(K))
        for i in range(n):
            poses[i][:] = d["init_poses"][i] + scale * rng.standard_normal(6)

    return d, host, group, blocks, plain, fill


def run_pass(host, state, blocks, plain, want_jac=True):
    intr, _, poses = host.views(state)
    for i, (b, q) in enumerate(zip(blocks, plain)):
        r, J = b.Evaluate([intr, poses[i]], want_jacobians=want_jac)
        r0, J0 = q.Evaluate([intr.copy(), poses[i].copy()], want_jacobians=want_jac)
        assert np.array_equal(r, r0), i
        if want_jac:
            assert np.array_equal(J[0], J0[0]) and np.array_equal(J[1], J0[1]), i


@pytest.mark.parametrize("model", ["eucm", "mei"])
def test_state_vector_host_is_served_from_one_pass_per_point(model):
    import visgeom_amd as vg

    n = 37
    d, host, group, blocks, plain, fill = make(vg, model, n, "state_vector")
    x, xpd = np.zeros(host.size), np.zeros(host.size)
    fill(x, 0.0)
    run_pass(host, x, blocks, plain)                       # pass 1: every block alone, bound, group sealed
    assert group.stats() == {"blocks": n, "batched": 0, "served": 0, "alone": n}
    fill(xpd, 1e-3)
    run_pass(host, xpd, blocks, plain)                     # pass 2, another array: which pointers travel is learned
    assert group.stats() == {"blocks": n, "batched": 0, "served": 0, "alone": 2 * n}
    fill(x, 2e-3)
    run_pass(host, x, blocks, plain, want_jac=False)       # candidate: cost only -> ONE pass serves everybody
    assert group.stats() == {"blocks": n, "batched": 1, "served": n, "alone": 2 * n}
    run_pass(host, x, blocks, plain)                       # accepted: Jacobians at the same point -> one more pass
    assert group.stats()["batched"] == 2 and group.stats()["alone"] == 2 * n
    run_pass(host, x, blocks, plain)                       # the same point again: nothing is recomputed
    assert group.stats()["batched"] == 2 and group.stats()["served"] == 3 * n
    fill(xpd, 3e-3)
    run_pass(host, xpd, blocks, plain)                     # the other array, new values
    fill(xpd, 4e-3)
    run_pass(host, xpd, blocks, plain)                     # new values in the SAME array (no displacement)
    s = group.stats()
    assert s["batched"] == 4 and s["alone"] == 2 * n and s["served"] == 5 * n
    for b in blocks + plain:
        b.close()
    group.close()


def test_an_array_the_group_was_never_shown_is_not_read_and_invalidate_forgets():
    """a displaced address is formed by pointer arithmetic; it is only READ when it lies inside memory some
    block has been called with.  A third state array therefore costs one block-by-block pass (each call shows its
    addresses) before it is batched; vg_block_group_invalidate (the host calls it when ceres::Solve returns and its state
    arrays are freed) voids everything the group knew about addresses."""
    import visgeom_amd as vg

    n = 21
    d, host, group, blocks, plain, fill = make(vg, "eucm", n, "state_vector")
    x, xpd = np.zeros(host.size), np.zeros(host.size)
    fill(x, 0.0)
    run_pass(host, x, blocks, plain)
    fill(xpd, 1e-3)
    run_pass(host, xpd, blocks, plain)
    fill(x, 2e-3)
    run_pass(host, x, blocks, plain)
    assert group.stats() == {"blocks": n, "batched": 1, "served": n, "alone": 2 * n}
    y = np.zeros(host.size)                                # a state array no block has ever been called with
    fill(y, 3e-3)
    run_pass(host, y, blocks, plain)                       # pass refused at its first call, every block alone
    assert group.stats() == {"blocks": n, "batched": 1, "served": n, "alone": 3 * n}
    fill(y, 4e-3)
    run_pass(host, y, blocks, plain)                       # now y has been shown: one pass again
    assert group.stats() == {"blocks": n, "batched": 2, "served": 2 * n, "alone": 3 * n}
    # the solve is over: state arrays go away, the report evaluates on user memory (one array per block)
    group.invalidate()
    del x, xpd, y
    user_intr = d["init_intrinsics"].copy()
    user_poses = [d["init_poses"][i].copy() for i in range(n)]

    def user_pass():
        for i, (b, q) in enumerate(zip(blocks, plain)):
            r, J = b.Evaluate([user_intr, user_poses[i]])
            r0, J0 = q.Evaluate([user_intr.copy(), user_poses[i].copy()])
            assert np.array_equal(r, r0) and np.array_equal(J[0], J0[0]) and np.array_equal(J[1], J0[1]), i

    user_pass()
    assert group.stats() == {"blocks": n, "batched": 2, "served": 2 * n, "alone": 4 * n}
    user_intr *= 1.001                                     # a new point, evaluated in place
    user_pass()
    st = group.stats()
    assert st["batched"] == 3 and st["alone"] == 4 * n and st["served"] == 3 * n
    for b in blocks + plain:
        b.close()
    group.close()


def test_constant_intrinsics_stay_in_user_memory():
    import visgeom_amd as vg

    n = 12
    d, host, group, blocks, plain, fill = make(vg, "ucm", n, "state_vector", constant_intrinsics=True)
    x, xpd = np.zeros(host.size), np.zeros(host.size)
    for k, (state, scale) in enumerate(((x, 0.0), (xpd, 1e-3), (x, 2e-3), (xpd, 1e-3), (x, 5e-3))):
        fill(state, scale)
        host.user_intr[:] = d["init_intrinsics"]
        run_pass(host, state, blocks, plain)
    s = group.stats()
    assert s["alone"] == 2 * n and s["batched"] == 3 and s["served"] == 3 * n
    for b in blocks + plain:
        b.close()
    group.close()


def test_wrong_mode_is_slow_not_wrong():
    """an in-place group under a host that moves its parameters: every prediction is stale, every block evaluates
    alone -- and still returns the right rows"""
    import visgeom_amd as vg

    n = 9
    d, host, group, blocks, plain, fill = make(vg, "eucm", n, "in_place")
    x, xpd = np.zeros(host.size), np.zeros(host.size)
    fill(x, 0.0)
    run_pass(host, x, blocks, plain)
    fill(xpd, 1e-3)
    run_pass(host, xpd, blocks, plain)
    s = group.stats()
    assert s["served"] == 1 and s["alone"] == 2 * n - 1        # one futile pass (its caller), then block by block
    fill(xpd, 2e-3)                 # ... and once the host does evaluate in place, the group catches up
    run_pass(host, xpd, blocks, plain)
    fill(xpd, 3e-3)
    run_pass(host, xpd, blocks, plain)
    assert group.stats()["served"] >= n + 1
    for b in blocks + plain:
        b.close()
    group.close()


def test_stereo_group_two_datasets_and_late_members():
    """cam-1 blocks [pose D] and cam-2 blocks [xi12 I, pose D] sharing the poses (data/calib_stereo_example.json): two
    datasets in the resident problem; a block added later unseals the group, which re-learns"""
    import visgeom_amd as vg
    from visgeom_amd import synthetic as S

    n = 10
    s = S.make_stereo(n)
    group = vg.BlockGroup(0, "state_vector")
    mk = lambda corners, status, g: vg.GenericProjectionJac(corners, s["board"], "eucm", status, group=g)
    b1 = [mk(s["corners1"][i], [0], group) for i in range(n - 1)]
    b2 = [mk(s["corners2"][i], [1, 0], group) for i in range(n - 1)]
    p1 = [mk(s["corners1"][i], [0], None) for i in range(n)]
    p2 = [mk(s["corners2"][i], [1, 0], None) for i in range(n)]
    size = 12 + 6 + 6 * n
    rng = np.random.default_rng(1)

    def fill(state, scale):
        state[0:6] = s["init_intrinsics1"] * (1 + scale * rng.standard_normal(6))
        state[6:12] = s["init_intrinsics2"] * (1 + scale * rng.standard_normal(6))
        state[12:18] = s["init_xi12"] + scale * rng.standard_normal(6)
        state[18:] = (s["init_poses"] + scale * rng.standard_normal((n, 6))).ravel()

    def run(state):
        i1, i2, x12 = state[0:6], state[6:12], state[12:18]
        for i in range(len(b1)):
            pose = state[18 + 6 * i:24 + 6 * i]
            r, J = b1[i].Evaluate([i1, pose])
            r0, J0 = p1[i].Evaluate([i1.copy(), pose.copy()])
            assert np.array_equal(r, r0) and all(np.array_equal(a, b) for a, b in zip(J, J0))
        for i in range(len(b2)):
            pose = state[18 + 6 * i:24 + 6 * i]
            r, J = b2[i].Evaluate([i2, x12, pose])
            r0, J0 = p2[i].Evaluate([i2.copy(), x12.copy(), pose.copy()])
            assert np.array_equal(r, r0) and all(np.array_equal(a, b) for a, b in zip(J, J0))

    x, xpd = np.zeros(size), np.zeros(size)
    fill(x, 0.0)
    run(x)
    fill(xpd, 1e-3)
    run(xpd)
    fill(x, 2e-3)
    run(x)
    st = group.stats()
    assert st["batched"] == 1 and st["served"] == 2 * (n - 1)
    b1.append(mk(s["corners1"][n - 1], [0], group))        # late members
    b2.append(mk(s["corners2"][n - 1], [1, 0], group))
    fill(xpd, 3e-3)
    run(xpd)                                               # the two new blocks are seen once: group sealed again
    fill(x, 4e-3)
    run(x)                                                 # ... and a second time: their pointers are known to travel
    fill(xpd, 5e-3)
    run(xpd)
    st2 = group.stats()
    assert st2["blocks"] == 2 * n and st2["batched"] == st["batched"] + 1 and st2["served"] == st["served"] + 2 * n
    # destroy order: group first, blocks continue alone
    group.close()
    run(x)
    for b in b1 + b2 + p1 + p2:
        b.close()
