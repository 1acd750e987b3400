import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visgeom_amd import CalibrationProblem, synthetic
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set
d = synthetic.make_mono("mei", 2000, 4)
p = CalibrationProblem(0)
cam = p.add_camera("mei", d["init_intrinsics"]); seq = p.add_transform(False, d["init_poses"])
p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"]); p.finalize()
s = p.solve(max_num_iterations=100, verbose=1)
print(s["termination"], s["num_iterations"], p.get_parameters()[:10])
