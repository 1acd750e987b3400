import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visgeom_amd import capi, synthetic as S
import visgeom_amd as vg
from tests.test_gpu_rig import build_rig
for n in (100, 400, 1500):
    r = S.make_rig(n, sigma=0.1)
    for mode in ("host", "device"):
        capi.debug_set("solver_device_loop", 1 if mode == "device" else 0)
        best = None
        for rep in range(4):
            p = build_rig(vg, r)[0]
            s = p.solve(max_num_iterations=200)
            p.close()
            if best is None or s["total_seconds"] < best["total_seconds"]:
                best = s
        print(n, mode, best["num_iterations"], "%.2f ms" % (best["total_seconds"] * 1e3), "%.3f ms/iter" % (best["total_seconds"] * 1e3 / best["num_iterations"]), best["termination"])
