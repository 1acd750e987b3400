"""one evaluation (chain + emit) as ONE launch (inline chain) vs TWO (chain-prep kernel + emit), by image count"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visgeom_amd import CalibrationProblem, synthetic
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set
n = int(sys.argv[1])
model = sys.argv[2] if len(sys.argv) > 2 else "eucm"
d = synthetic.make_mono(model, n, 1)
p = CalibrationProblem(0)
cam = p.add_camera(model, d["init_intrinsics"]); seq = p.add_transform(False, d["init_poses"])
ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"]); p.finalize()
res, ji, jm = p.alloc_outputs(ds)
def step():
    p.prepare(); p.evaluate_dataset(ds, res, ji, jm)
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:   # steady clocks (see step_overhead_probe.py)
    for _ in range(50): step()
    torch.cuda.synchronize()
for _ in range(30): step()
reps = max(50, 10000000 // n)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): step()
e1.record(); torch.cuda.synchronize()
print(model, "images %7d  VG_INLINE_CHAIN_MAX_BYTES=%s  step %.1f us" % (n, os.environ.get("VG_INLINE_CHAIN_MAX_BYTES", "default"), e0.elapsed_time(e1) / reps * 1e3))
