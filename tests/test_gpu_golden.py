"""The HIP solver against the committed scipy-on-oracle optima of BASELINE.json configs 2-5 (reduced image count):
"converge to the same intrinsics within 1e-6" (north_star) anchored on every camera model, the stereo chain and the
four-camera rig -- tests/golden/optimum_*.json, generator tools/gen_golden.py, cases tests/golden_cases.py."""
import json
import os

import numpy as np
import pytest

from tests import golden_cases as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", G.NAMES)
def test_solver_reaches_the_committed_optimum(name):
    import visgeom_amd as vg

    with open(os.path.join(ROOT, "tests", "golden", "optimum_%s.json" % name)) as f:
        fx = json.load(f)
    c = G.case(name)
    assert G.input_digest(c) == fx["input_digest"]
    p = G.build_product_problem(vg, c)
    assert p.num_parameters == fx["n_parameters"]
    s = p.solve(max_num_iterations=300)
    x = p.get_parameters()
    ref = np.array(fx["x_opt"])
    ng = fx["n_global_parameters"]  # intrinsics of every camera + the global transforms
    rel_glob = np.max(np.abs(x[:ng] - ref[:ng]) / np.maximum(np.abs(ref[:ng]), 1.0))
    rel_pose = np.max(np.abs(x[ng:] - ref[ng:]))
    print(name, s["termination"], s["num_iterations"], "cost %.12e (golden %.12e) globals %.2e poses %.2e" %
          (s["final_cost"], fx["cost"], rel_glob, rel_pose))
    assert abs(s["final_cost"] - fx["cost"]) <= 1e-9 * fx["cost"]
    assert rel_glob <= 1e-6
    assert rel_pose <= 1e-6
    p.close()
