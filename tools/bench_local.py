#!/usr/bin/env python3
"""One-GPU measurement of the localization reprojection costs (visgeom_amd/csrc/vg_local.hpp):
evaluations of resident block sets through the batched device entries, HIP-event timings, algorithmic bytes per feature and
the HBM fraction they amount to.  tools only -- bench.py stays the driver's contract (the calibration metric).

  ransac   SparseReprojectCost, 200 hypotheses x 8 points   (SparseOdometry::ransacNPoints, sparse_odom.cpp:511-606)
  inliers  SparseReprojectCost, 1 block x 2 000 points      (the final computeTransfSparse on all inliers, :384)
  large    SparseReprojectCost, 2 000 blocks x 500 points = 1 M features
  mono     MonoReprojectCost, 200 000 five-point blocks = 1 M features

Algorithmic bytes per feature (what one evaluation has to move): sparse 48 (x1, x2) + 16 (p2) + 8 (size) + 4 (block index)
read, 16 (residual pair) + 96 (2 x 6 Jacobian) written = 188 B; mono 24 + 16 + 8 (length) read, 16 + 96 + 80 (2 x 5 length
rows) written = 240 B (+ 9.6 B of the block's odometry parameter).  The per-block frame (70 / 34 doubles) is computed in the
point kernel for mono sets and for sparse launches of at most 256 workgroups; large sparse sets amortise a frame launch.

usage: python tools/bench_local.py [reps]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from visgeom_amd import _build  # noqa: E402

if os.environ.get("AB_LIB"):   # same-box A/B against a variant library (python -m visgeom_amd._build --variant NAME ...)
    _build.LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["AB_LIB"])
from visgeom_amd import localization as loc  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
HBM_PEAK = 8.0e12
INTR = {"eucm": [0.571, 1.18, 312.0, 305.0, 655.0, 391.0], "mei": [1.27, -0.04, 0.012, -0.003, 0.0012, -0.0017, 698.0, 705.0, 648.0, 395.0]}
XB = np.array([0.21, -0.08, 0.33, 0.12, -1.15, 1.07])


def rotmat(r):
    th = np.linalg.norm(r)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def T(x):
    A = np.eye(4)
    A[:3, :3], A[:3, 3] = rotmat(x[3:]), x[:3]
    return A


def scene(rng, n, xo):
    X1 = np.column_stack([rng.uniform(-1, 1, n), rng.uniform(-0.7, 0.7, n), rng.uniform(2, 6, n)])
    x1 = X1 / np.linalg.norm(X1, axis=1)[:, None]
    T12 = np.linalg.inv(T(XB)) @ T(xo) @ T(XB)
    X2 = (np.linalg.inv(T12) @ np.c_[X1, np.ones(n)].T).T[:, :3]
    x2 = X2 / np.linalg.norm(X2, axis=1)[:, None] + 1e-3 * rng.standard_normal((n, 3))
    return X1, x1, x2, rng.uniform(300, 900, (n, 2)), rng.uniform(1, 4, n)


def timed(fn, reps=REPS):
    for _ in range(max(3, reps // 10)):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    rng = np.random.default_rng(7)
    xo = np.array([0.31, 0.04, -0.02, 0.012, -0.021, 0.083])
    rows = []
    _, x1, x2, p2, size = scene(rng, 2000, xo)
    cases = [("ransac: 200 hypotheses x 8 points", "eucm", True, [tuple(a[i] for a in (x1, x2, p2, size)) for i in
                                                                  (rng.permutation(2000)[:8] for _ in range(200))]),
             ("inliers: 1 block x 2000 points", "eucm", True, [(x1, x2, p2, size)]),
             ("large: 2000 blocks x 500 points", "mei", True, [(x1[:500], x2[:500], p2[:500], size[:500])] * 2000)]
    for name, model, sparse, blocks in cases:
        st = loc.ReprojectSet(model, INTR[model], XB, blocks, sparse=sparse)
        xos = torch.tensor(xo + 0.01 * rng.standard_normal((len(blocks), 6)), device="cuda")
        t = timed(lambda: st.evaluate(xos))
        t_res = timed(lambda: st.evaluate(xos, want_jac=False))
        n = st.n_points
        rows.append({"case": name, "model": model, "features": n, "blocks": len(blocks), "evaluation_us": t * 1e6, "cost_only_us": t_res * 1e6,
                     "features_per_s": n / t, "algorithmic_bytes": 188 * n, "GBps": 188 * n / t / 1e9, "frac_of_hbm_peak": 188 * n / t / HBM_PEAK})
        st.close()
    nb = 200000
    X1, x1m, _, p2m, _ = scene(rng, 5, xo)
    st = loc.ReprojectSet("eucm", INTR["eucm"], XB, [(x1m, p2m)] * nb, sparse=False)
    xos = torch.tensor(xo + 0.01 * rng.standard_normal((nb, 6)), device="cuda")
    lens = torch.tensor(np.tile(np.linalg.norm(X1, axis=1), (nb, 1)), device="cuda")
    t = timed(lambda: st.evaluate(xos, lens), max(20, REPS // 4))
    rows.append({"case": "mono: 200000 five-point blocks", "model": "eucm", "features": 5 * nb, "blocks": nb, "evaluation_us": t * 1e6,
                 "features_per_s": 5 * nb / t, "algorithmic_bytes": 240 * 5 * nb, "GBps": 240 * 5 * nb / t / 1e9,
                 "frac_of_hbm_peak": 240 * 5 * nb / t / HBM_PEAK})
    st.close()
    # CameraJacobian::dpdxi / dfdxi (jacobian.h:51-119), two-transform constructor, 1 M points of a depth map: 24 (X2) + 16
    # (gradient) read, 96 (dpdxi) + 48 (dfdxi) written = 184 B per point
    import ctypes
    from visgeom_amd import capi
    L = capi.load()
    n = 1000000
    X = torch.tensor(np.column_stack([rng.uniform(-1, 1, n), rng.uniform(-0.7, 0.7, n), rng.uniform(1.5, 5, n)]), device="cuda")
    g = torch.tensor(rng.standard_normal((n, 2)), device="cuda")
    dp = torch.empty((n, 2, 6), dtype=torch.float64, device="cuda")
    df = torch.empty((n, 6), dtype=torch.float64, device="cuda")
    T12, T23 = np.array([0.3, -0.2, 0.1, 0.4, -0.3, 0.2]), np.array([-0.1, 0.25, 0.05, -0.2, 0.1, 0.5])
    intr = np.ascontiguousarray(INTR["mei"], float)
    dpt = ctypes.POINTER(ctypes.c_double)
    st_ptr = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    def cj():
        capi.check(L.vg_camera_jacobian_evaluate(0, st_ptr, 2, intr.ctypes.data_as(dpt), T12.ctypes.data_as(dpt), T23.ctypes.data_as(dpt), n, vp(X), vp(g), vp(dp), vp(df)))
    for _ in range(400):   # a one-time ~40 ms host stall sits somewhere in the first ~200 calls of this entry on the boxes used
        cj()
    torch.cuda.synchronize()
    t = timed(cj)
    rows.append({"case": "camera jacobian: 1 M points, two transforms", "model": "mei", "features": n, "blocks": 1, "evaluation_us": t * 1e6,
                 "features_per_s": n / t, "algorithmic_bytes": 184 * n, "GBps": 184 * n / t / 1e9, "frac_of_hbm_peak": 184 * n / t / HBM_PEAK})
    for r in rows:
        print(json.dumps(r))
    print()
    print("| case | model | features | evaluation us (all launches) | features/s | algorithmic GB/s | fraction of 8 TB/s |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %s | %d | %.1f | %.3e | %.0f | %.3f |" % (r["case"], r["model"], r["features"], r["evaluation_us"], r["features_per_s"], r["GBps"],
                                                             r["frac_of_hbm_peak"]))


if __name__ == "__main__":
    main()
