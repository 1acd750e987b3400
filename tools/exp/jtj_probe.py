import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visgeom_amd import CalibrationProblem, synthetic
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set
model = sys.argv[1] if len(sys.argv) > 1 else "eucm"
d = synthetic.make_mono(model, 10000, 1)
p = CalibrationProblem(0)
cam = p.add_camera(model, d["init_intrinsics"]); seq = p.add_transform(False, d["init_poses"])
ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"]); p.finalize()
gram, gsum = p.alloc_gram(ds)
p.prepare()
for _ in range(20): p.gram_fused(ds, gram)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): p.gram_fused(ds, gram)
e1.record(); torch.cuda.synchronize()
print(model, "VG_GRAM_WAVES", os.environ.get("VG_GRAM_WAVES"), "fused kernel us", e0.elapsed_time(e1) / 200 * 1e3)
