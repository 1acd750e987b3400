import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visgeom_amd import CalibrationProblem, synthetic
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set
for model in ("mei", "eucm"):
  for cfg in (1, 4):
    d = synthetic.make_mono(model, 10000, cfg)
    p = CalibrationProblem(0)
    cam = p.add_camera(model, d["init_intrinsics"]); seq = p.add_transform(False, d["init_poses"])
    ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"]); p.finalize()
    res, ji, jm = p.alloc_outputs(ds)
    p.prepare()
    for rep in range(3):
        for _ in range(50): p.evaluate_dataset(ds, res, ji, jm)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): p.evaluate_dataset(ds, res, ji, jm)
        e1.record(); torch.cuda.synchronize()
        print(model, "cfg", cfg, "rep", rep, "emit us %.2f" % (e0.elapsed_time(e1) / 300 * 1e3), "failed", p.failed_count(ds))
    p.close()
