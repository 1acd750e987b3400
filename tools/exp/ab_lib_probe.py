"""A/B of two builds of the library on the same box: AB_LIB=<path to alternative .so> python tools/exp/ab_lib_probe.py"""
import os, sys, time, torch
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
from visgeom_amd import _build
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set
if os.environ.get("AB_LIB"):
    _build.LIB = os.path.join(root, os.environ["AB_LIB"])
from visgeom_amd import CalibrationProblem, synthetic
d = synthetic.make_mono("eucm", 10000, 1)
p = CalibrationProblem(0)
cam = p.add_camera("eucm", d["init_intrinsics"]); seq = p.add_transform(False, d["init_poses"])
ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"]); p.finalize()
res, ji, jm = p.alloc_outputs(ds)
def step():
    p.prepare(); p.evaluate_dataset(ds, res, ji, jm)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:
    for _ in range(50): step()
    torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000): step()
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("AB_LIB", "default"), "step %.2f us" % (e0.elapsed_time(e1)))
