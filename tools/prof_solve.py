#!/usr/bin/env python3
"""profiling driver: full LM loop on config 5 (rig), config 3 (stereo) or a mono set (for rocprofv3).
usage: python tools/prof_solve.py <rig|stereo|eucm|ucm|mei> [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from visgeom_amd import _build as _b  # noqa: E402

if os.environ.get("AB_LIB"):   # same-box A/B against a variant library
    _b.LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["AB_LIB"])
from visgeom_amd import CalibrationProblem, synthetic  # noqa: E402
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set

which = sys.argv[1] if len(sys.argv) > 1 else "rig"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
if which == "rig":
    r = synthetic.make_rig(n)
    p = CalibrationProblem(0)
    cams = [p.add_camera(m, r["init_intrinsics"][k]) for k, m in enumerate(r["models"])]
    x1k = [p.add_transform(True, r["init_xi1k"][k]) for k in range(3)]
    seq = p.add_transform(False, r["init_poses"])
    p.add_dataset(cams[0], [(seq, 0)], r["board"], r["corners"][0])
    for k in range(3):
        p.add_dataset(cams[k + 1], [(x1k[k], 1), (seq, 0)], r["board"], r["corners"][k + 1])
elif which == "stereo":
    st = synthetic.make_stereo(n)
    p = CalibrationProblem(0)
    c1 = p.add_camera("eucm", st["init_intrinsics1"])
    c2 = p.add_camera("eucm", st["init_intrinsics2"])
    x12 = p.add_transform(True, st["init_xi12"])
    seq = p.add_transform(False, st["init_poses"])
    p.add_dataset(c1, [(seq, 0)], st["board"], st["corners1"])
    p.add_dataset(c2, [(x12, 1), (seq, 0)], st["board"], st["corners2"])
else:
    d = synthetic.make_mono(which, n, 1)
    p = CalibrationProblem(0)
    cam = p.add_camera(which, d["init_intrinsics"])
    seq = p.add_transform(False, d["init_poses"])
    p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"])
p.finalize()
x0 = p.get_parameters()
for rep in range(int(os.environ.get('REPS', '3'))):
    p.set_parameters(x0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    s = p.solve(max_num_iterations=150)
    print(rep, s["termination"], s["num_iterations"], "total %.2f ms  eval %.2f  schur %.2f  host %.2f  wall %.2f" % (
        s["total_seconds"] * 1e3, s["evaluate_seconds"] * 1e3, s["schur_seconds"] * 1e3, s["host_seconds"] * 1e3,
        (time.perf_counter() - t) * 1e3))
p.close()
