import os, sys
sys.path.insert(0, "/root/repo")
from visgeom_amd import synthetic as S
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set
from visgeom_amd.problem import CalibrationProblem
model = sys.argv[1]; n = int(sys.argv[2])
d = S.make_mono(model, n, 1 if model == "eucm" else 4)
p = CalibrationProblem(0)
c = p.add_camera(model, d["init_intrinsics"]); s = p.add_transform(False, d["init_poses"])
p.add_dataset(c, [(s, 0)], d["board"], d["corners"]); p.finalize()
summ = p.solve(max_num_iterations=100, verbose=1)
print(summ["num_iterations"], summ["num_successful_steps"], summ["termination"], summ["final_cost"])
