"""CPU side of the committed optima (tests/golden/optimum_*.json, written by tools/gen_golden.py): every stored vector
belongs to the inputs the seeded generator produces today, reproduces its stored cost on the oracle, and is a
stationary point of the oracle's least-squares cost (projected onto the cameras' boxes)."""
import json
import os

import numpy as np
import pytest

from tests import golden_cases as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    with open(os.path.join(ROOT, "tests", "golden", "optimum_%s.json" % name)) as f:
        return json.load(f)


@pytest.mark.parametrize("name", G.NAMES)
def test_stored_optimum_is_a_stationary_point_of_the_oracle_cost(name):
    fx = load(name)
    c = G.case(name)
    assert G.input_digest(c) == fx["input_digest"], "the seeded generator no longer produces the fixture's inputs"
    cam_off, tf_off, x0, lb, ub = G.layout(c)
    assert (cam_off, tf_off, x0.size) == (fx["camera_offsets"], fx["transform_offsets"], fx["n_parameters"])
    x = np.array(fx["x_opt"])
    r, J = G.oracle_rows(c, x)
    assert r.size == fx["n_residuals"]
    cost = 0.5 * r @ r
    assert abs(cost - fx["cost"]) <= 1e-12 * fx["cost"]
    assert cost < 1e-3 * fx["initial_cost"]
    g = J.T @ r
    free = (x > lb) & (x < ub)
    # first-order optimality, column by column: |J_j^T r| against |J_j| |r| (the cosine of the angle between the
    # residual and the column); parameters on a bound must push outwards
    cn = np.linalg.norm(J, axis=0)
    assert np.all(cn > 0)
    assert np.max(np.abs(g[free]) / (cn[free] * np.linalg.norm(r))) < 1e-9
    for i in np.nonzero(~free)[0]:
        assert (x[i] <= lb[i] and g[i] >= 0) or (x[i] >= ub[i] and g[i] <= 0)
    # and it is a minimum along random directions (second-order sanity)
    rng = np.random.default_rng(5)
    for _ in range(3):
        d = rng.standard_normal(x.size) * 1e-6 * np.maximum(np.abs(x), 1.0)
        d[~free] = 0.0
        rp = G.oracle_rows(c, x + d, want_jac=False)[0]
        assert 0.5 * rp @ rp >= cost * (1 - 1e-13)
