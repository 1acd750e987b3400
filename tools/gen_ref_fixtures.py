#!/usr/bin/env python3
"""Golden vectors from the REFERENCE's own GenericProjectionJac::Evaluate (src/calibration/calib_cost_functions.cpp:28-117),
through oracle/_ref/libvg_ref.so (oracle/build_ref.sh: real Eigen3 + Ceres, no stand-ins).  Writes
tests/golden/ref_eval_block.json = inputs + the reference's residuals and Jacobians for the SURVEY section 7 step 1 case
matrix; tests/test_oracle_vs_ref_fixtures.py then pins oracle/vg_oracle.c to them (and skips while the file is absent).

  3 camera models  x  chains {[D], [I, D], [D, I, D], [I, D, D, I, D]}  x  branch edges:
    generic pose | rot = 0 (first-order branches of Quaternion / rotationMatrix / interOmegaRot) | |rot| at 1e-6, 1e-5 +- 1 ulp
    | |rot| > pi (toRotationVector wrap) | board behind the camera (EUCM: in-band 1e15 rows) | NULL Jacobian patterns

usage: bash oracle/build_ref.sh && python tools/gen_ref_fixtures.py        (exits 77 when the library is absent)
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libvg_ref.so")
OUT = os.path.join(ROOT, "tests", "golden", "ref_eval_block.json")
_dp = ctypes.POINTER(ctypes.c_double)

INTR = {0: [0.595728, 0.768828, 307.318, 289.542, 642.617, 398.42],
        1: [1.2, 307.318, 289.542, 642.617, 398.42],
        2: [1.2, -0.05, 0.01, -0.002, 0.001, -0.0015, 307.318, 289.542, 642.617, 398.42]}
CHAINS = [[0], [1, 0], [0, 1, 0], [1, 0, 0, 1, 0]]
BOARD_POSE = [-0.35, -0.2, 0.9, 0.3, -0.4, 0.1]


def cases():
    """deterministic inputs (no RNG state shared with anything else)"""
    rng = np.random.default_rng(20260929)
    board = np.array([[0.1 * j, 0.1 * i, 0.0] for i in range(3) for j in range(4)])   # 12 points: small fixtures
    out = []
    for model in (0, 1, 2):
        for status in CHAINS:
            L = len(status)
            edges = [("generic", None), ("rot0", 0.0), ("rot1e-6", 1e-6), ("rot1e-6+ulp", np.nextafter(1e-6, 1)),
                     ("rot1e-5", 1e-5), ("rot1e-5-ulp", np.nextafter(1e-5, 0)), ("rot>pi", 3.5), ("behind", None)]
            for name, rot in edges:
                members = []
                for l in range(L):
                    xi = np.array(BOARD_POSE) if l == L - 1 else np.concatenate([rng.uniform(-0.05, 0.05, 3), rng.uniform(-0.1, 0.1, 3)])
                    if rot is not None and l == L - 1:
                        u = np.array([0.3, -0.4, 0.1])
                        xi[3:] = u / np.linalg.norm(u) * rot
                    if name == "behind" and l == L - 1:
                        xi[:3] = [0.0, 0.0, -1.0]
                        xi[3:] = 0.0
                    members.append(xi)
                obs = rng.uniform(200, 900, (board.shape[0], 2))
                for mask_name, mask in (("all", [True] * (L + 1)), ("intr_const", [False] + [True] * L)):
                    if mask_name != "all" and name != "generic":
                        continue
                    out.append({"name": "m%d_%s_%s_%s" % (model, "".join("DI"[s] for s in status), name, mask_name), "model": model,
                                "status": status, "board": board.tolist(), "obs": obs.tolist(), "intrinsics": INTR[model],
                                "members": [m.tolist() for m in members], "jac_mask": mask})
    return out


def main():
    if not os.path.exists(LIB):
        print("gen_ref_fixtures: %s is absent (oracle/build_ref.sh needs real Eigen3 + Ceres)" % LIB)
        sys.exit(77)
    lib = ctypes.CDLL(LIB)
    lib.ref_eval_block.restype = ctypes.c_int
    done = []
    for c in cases():
        L, N = len(c["status"]), len(c["board"])
        K = len(c["intrinsics"])
        ps = [np.array(c["intrinsics"], float)] + [np.array(m, float) for m in c["members"]]
        pp = (_dp * (L + 1))(*[p.ctypes.data_as(_dp) for p in ps])
        st = (ctypes.c_int * L)(*c["status"])
        grid, obs = np.ascontiguousarray(c["board"], float), np.ascontiguousarray(c["obs"], float)
        res = np.empty(2 * N)
        jacs = [np.full((2 * N, s), np.nan) for s in [K] + [6] * L]
        jp = (_dp * (L + 1))(*[j.ctypes.data_as(_dp) if m else _dp() for j, m in zip(jacs, c["jac_mask"])])
        ok = lib.ref_eval_block(c["model"], L, st, N, grid.ctypes.data_as(_dp), obs.ctypes.data_as(_dp), pp, res.ctypes.data_as(_dp), jp)
        c["ref_return"] = int(ok)
        c["ref_residual"] = [float.hex(float(v)) for v in res]
        c["ref_jacobians"] = [[float.hex(float(v)) for v in j.ravel()] if m else None for j, m in zip(jacs, c["jac_mask"])]
        done.append(c)
    with open(OUT, "w") as fh:
        json.dump({"generator": "tools/gen_ref_fixtures.py", "library": "oracle/_ref/libvg_ref.so (oracle/build_ref.sh)",
                   "reference": "src/calibration/calib_cost_functions.cpp:28-117", "cases": done}, fh, indent=0)
    print("wrote %d cases -> %s" % (len(done), OUT))


if __name__ == "__main__":
    main()
