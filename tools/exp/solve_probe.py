"""Warm timing of vg_problem_solve: set-up vs iterations, with / without speculative queueing (VG_SOLVER_NO_SPECULATION=1).
usage: python tools/exp/solve_probe.py [model] [images] [repeats]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np

from visgeom_amd import synthetic as S
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set
if os.environ.get("AB_LIB"):   # an A/B library (python -m visgeom_amd._build --variant NAME -DFLAG)
    from visgeom_amd import _build
    _build.LIB = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), os.environ["AB_LIB"])
from visgeom_amd.problem import CalibrationProblem

model = sys.argv[1] if len(sys.argv) > 1 else "eucm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
d = S.make_mono(model, n, 1 if model == "eucm" else 4)
import torch

side = torch.cuda.Stream() if os.environ.get("VG_PROBE_OWN_STREAM") else None  # a capturable (non-NULL) stream
for r in range(reps):
    p = CalibrationProblem(0, stream=side.cuda_stream if side is not None else None)
    c = p.add_camera(model, d["init_intrinsics"])
    s = p.add_transform(False, d["init_poses"])
    p.add_dataset(c, [(s, 0)], d["board"], d["corners"])
    p.finalize()
    summ = p.solve(max_num_iterations=100)
    print("%s n=%d run %d: %d it (%d ok) %s total %.3f ms = set-up %.3f + loop %.3f (%.4f ms/it) cost %.6e" % (
        model, n, r, summ["num_iterations"], summ["num_successful_steps"], summ["termination"], summ["total_seconds"] * 1e3,
        summ["host_seconds"] * 1e3, summ["evaluate_seconds"] * 1e3, summ["evaluate_seconds"] * 1e3 / max(1, summ["num_iterations"]),
        summ["final_cost"]), flush=True)
    p.close()
