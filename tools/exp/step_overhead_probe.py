import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visgeom_amd import CalibrationProblem, synthetic
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set
d = synthetic.make_mono("eucm", 10000, 1)
p = CalibrationProblem(0)
cam = p.add_camera("eucm", d["init_intrinsics"]); seq = p.add_transform(False, d["init_poses"])
ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"]); p.finalize()
res, ji, jm = p.alloc_outputs(ds)
def step():
    p.prepare(); p.evaluate_dataset(ds, res, ji, jm)
for _ in range(20): step()
torch.cuda.synchronize()
for K in (20, 50, 200, 1000, 5000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("K %5d  enqueue %.1f us/step   total %.2f us/step" % (K, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))

# GPU-side time of consecutive windows of one long run (events every 50 steps)
torch.cuda.synchronize()
time.sleep(0.5)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
evs[0].record()
for w in range(40):
    for _ in range(50): step()
    evs[w + 1].record()
torch.cuda.synchronize()
print("us/step per 50-step window:", " ".join("%.1f" % (evs[i].elapsed_time(evs[i + 1]) / 50 * 1e3) for i in range(40)))
