"""CPU check of tests/oracle_gn.py itself (the helper the full-size GPU tests lean on): its arrow-structured
Gauss-Newton step equals a dense least-squares step of the oracle's stacked Jacobian, at the committed optima (where it
must vanish) and away from them."""
import json
import os

import numpy as np
import pytest

from tests import golden_cases as G
from tests import oracle_gn as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scatter(g, n):
    full = np.zeros(n)
    full[g["gcols"]] = g["dg"]
    for k, pp in enumerate(g["pose_param"]):
        full[pp:pp + 6] = g["dp"][k]
    return full


@pytest.mark.parametrize("name", G.NAMES)
def test_arrow_step_equals_the_dense_least_squares_step(name):
    with open(os.path.join(ROOT, "tests", "golden", "optimum_%s.json" % name)) as f:
        fx = json.load(f)
    c = G.case(name)
    x = np.array(fx["x_opt"])
    g = O.gauss_newton_step(c, x)
    assert abs(g["cost"] - fx["cost"]) <= 1e-12 * fx["cost"]
    O.assert_converged(c, x, fx["cost"], tol=1e-6, what=name)     # a committed optimum passes the full-size bar
    for xs in (x, x * (1 + 1e-4)):
        gs = O.gauss_newton_step(c, xs)
        r, J = G.oracle_rows(c, xs)
        d = np.linalg.lstsq(J, -r, rcond=None)[0]
        assert np.max(np.abs(scatter(gs, x.size) - d)) <= 1e-6 * max(np.max(np.abs(d)), 1e-3)
    # ... and a point that is NOT converged fails it
    with pytest.raises(AssertionError):
        O.assert_converged(c, x * (1 + 1e-4), fx["cost"], tol=1e-6)


def test_shard_case_partitions_the_images():
    c = G.case("stereo")
    a, b = O.shard_case(c, 0, 17), O.shard_case(c, 17, 40)
    for (_, _, _, ca), (_, _, _, cb), (_, _, _, cc) in zip(a["datasets"], b["datasets"], c["datasets"]):
        assert np.array_equal(np.concatenate([ca, cb]), cc)
    assert a["transforms"][0][1].shape == (1, 6) and a["transforms"][1][1].shape == (17, 6)
