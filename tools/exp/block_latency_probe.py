"""latency of the per-block drop-in entry (vg_block_evaluate): H2D params, two launches, one D2H, one synchronisation"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from visgeom_amd import GenericProjectionJac, synthetic
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set
d = synthetic.make_mono("eucm", 4, 2)
cases = (([0], [d["init_intrinsics"], d["init_poses"][0]]),
         ([1, 0], [d["init_intrinsics"], np.array([0.1, 0, 0, 0.01, 0, 0.0]), d["init_poses"][0]]))
blks = [GenericProjectionJac(d["corners"][0], d["board"], "eucm", st) for st, _ in cases]
for rep in range(3):
    for (status, params), blk in zip(cases, blks):
        for _ in range(50): blk.Evaluate(params)
        n = 500
        t0 = time.perf_counter()
        for _ in range(n): blk.Evaluate(params)
        t1 = time.perf_counter()
        print("rep %d chain %-6s per-block Evaluate with all Jacobians: %.1f us" % (rep, status, (t1 - t0) / n * 1e6))
