#!/usr/bin/env python3
"""Writes tests/golden/optimum_<case>.json: the least-squares optimum of BASELINE.json configs 2-5 at a reduced image
count, found with scipy.optimize.least_squares(method="trf", bounds = the cameras' boxes, x_scale="jac",
xtol = ftol = gtol = 1e-15) over the ORACLE's residuals and Jacobians (oracle/vg_oracle.c).  The optimum of a
least-squares problem is a property of the problem, not of the solver -- these vectors stand in for "what Ceres would
converge to" (SURVEY 8(c)); the reference itself cannot be built here (no Eigen3 / Ceres in the image).

Run in the build container (CPU only, a few minutes):   python tools/gen_golden.py [case ...]
Inputs come from the seeded generator visgeom_amd/synthetic.py, so a fixture stores the answer plus a digest of the
inputs it belongs to; tests/golden_cases.py describes the problems for the generator and for the tests alike.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from scipy.optimize import least_squares

    from tests import golden_cases as G

    names = sys.argv[1:] or G.NAMES
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    for name in names:
        c = G.case(name)
        cam_off, tf_off, x0, lb, ub = G.layout(c)
        t0 = time.time()
        best = None
        x = x0
        # trf stops on its own tolerances; restart from its answer until the cost stops moving (tight optimum)
        for rounds in range(6):
            sol = least_squares(lambda v: G.oracle_rows(c, v, False)[0], x, jac=lambda v: G.oracle_rows(c, v)[1],
                                bounds=(lb, ub), method="trf", x_scale="jac", xtol=1e-15, ftol=1e-15, gtol=1e-15,
                                max_nfev=300)
            if best is not None and abs(best.cost - sol.cost) <= 1e-15 * sol.cost and np.max(np.abs(best.x - sol.x)) < 1e-12:
                best = sol
                break
            best, x = sol, sol.x
        r, J = G.oracle_rows(c, best.x)
        g = J.T @ r
        free = (best.x > lb) & (best.x < ub)
        n_glob = tf_off[-1] if not c["transforms"][-1][0] else x0.size  # everything in front of the (last) sequence
        out = {
            "case": name,
            "generator": "tests/golden_cases.py::case(%r) on visgeom_amd/synthetic.py" % name,
            "method": "scipy %s least_squares trf, x_scale=jac, tolerances 1e-15, %d restart round(s), oracle rows" %
                      (__import__("scipy").__version__, rounds + 1),
            "input_digest": G.input_digest(c),
            "n_parameters": int(x0.size), "n_residuals": int(r.size), "n_global_parameters": int(n_glob),
            "cost": float(0.5 * r @ r), "initial_cost": float(0.5 * np.sum(G.oracle_rows(c, x0, False)[0] ** 2)),
            "gradient_max_norm_free": float(np.max(np.abs(g[free]))) if free.any() else 0.0,
            "on_bound": [int(i) for i in np.nonzero(~free)[0]],
            "x_opt": [float(v) for v in best.x],
            "camera_offsets": cam_off, "transform_offsets": tf_off,
        }
        path = os.path.join(ROOT, "tests", "golden", "optimum_%s.json" % name)
        with open(path, "w") as f:
            json.dump(out, f, indent=0)
            f.write("\n")
        print("%-10s n=%d m=%d cost %.12e |g|max %.2e nfev %d %.0fs -> %s" % (name, x0.size, r.size, out["cost"],
              out["gradient_max_norm_free"], best.nfev, time.time() - t0, os.path.relpath(path, ROOT)))


if __name__ == "__main__":
    main()
