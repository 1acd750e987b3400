"""Many solves in a row through both LM loops (sequence-word waits, speculative queueing): every one must end at the same
cost in about the same time.  usage: python tools/exp/solve_stress.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np  # noqa: E402

import visgeom_amd as vg  # noqa: E402
from tests.test_gpu_rig import build_rig  # noqa: E402
from visgeom_amd import synthetic as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
r = S.make_rig(200, sigma=0.1)
d = S.make_mono("eucm", 500, 3)
for name in ("rig (host loop)", "mono (device loop)"):
    costs, times = [], []
    for k in range(n):
        if name.startswith("rig"):
            p = build_rig(vg, r)[0]
        else:
            p = vg.CalibrationProblem(0)
            cam = p.add_camera("eucm", d["init_intrinsics"])
            seq = p.add_transform(False, d["init_poses"])
            p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"])
            p.finalize()
        t = time.perf_counter()
        s = p.solve(max_num_iterations=200)
        times.append(time.perf_counter() - t)
        costs.append(s["final_cost"])
        assert s["termination"].startswith("CONVERGENCE"), s
        p.close()
    times = np.array(times[5:]) * 1e3
    print("%s: %d solves, cost spread %.1e, ms min / median / max %.2f / %.2f / %.2f" % (
        name, n, (max(costs) - min(costs)) / costs[0], times.min(), np.median(times), times.max()))
