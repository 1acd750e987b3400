"""The rig's LM iteration (config 5: 4 cameras x 5 000 frames, 45 global columns): host-driven loop vs the device-resident
loop forced through the debug hook, with the host loop's own split (evaluations / pose elimination / host algebra).
usage: python tools/exp/rig_loop_probe.py"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visgeom_amd import capi, synthetic as S  # noqa: E402
import visgeom_amd as vg  # noqa: E402
from tests.test_gpu_rig import build_rig  # noqa: E402

capi.hooks_from_env()   # VG_SOLVER_NO_FOLD_FRAMES=1 etc.
r = S.make_rig(5000, sigma=0.1)
for mode in ("host", "device"):
    capi.debug_set("solver_device_loop", 1 if mode == "device" else 0)
    best = None
    for rep in range(3):
        p, cams, x1k, seq, dss = build_rig(vg, r)
        s = p.solve(max_num_iterations=200)
        p.close()
        if best is None or s["total_seconds"] < best["total_seconds"]:
            best = s
    n = best["num_iterations"]
    print(mode, n, "%.2f ms" % (best["total_seconds"] * 1e3), "%.3f ms/iter" % (best["total_seconds"] * 1e3 / n), best["termination"],
          best["final_cost"], "per iteration us: evaluate %.1f schur %.1f host %.1f" % tuple(best[k] * 1e6 / n for k in
                                                                                           ("evaluate_seconds", "schur_seconds", "host_seconds")))
