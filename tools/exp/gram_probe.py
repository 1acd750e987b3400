"""Fused Gram kernels, same box: VG_GRAM_FORCE_MFMA=1 selects the matrix-core kernel for narrow blocks.
AB_LIB=<alternative .so> python tools/exp/gram_probe.py [model] [images]"""
import os, sys, torch, numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
from visgeom_amd import _build
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set
if os.environ.get("AB_LIB"):
    _build.LIB = os.path.join(root, os.environ["AB_LIB"])
from visgeom_amd import CalibrationProblem, synthetic
model = sys.argv[1] if len(sys.argv) > 1 else "eucm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
d = synthetic.make_mono(model, n, 1)
p = CalibrationProblem(0)
cam = p.add_camera(model, d["init_intrinsics"]); seq = p.add_transform(False, d["init_poses"])
ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"]); p.finalize()
gram, gsum = p.alloc_gram(ds)
gsum2 = torch.empty_like(gsum)
def t(fn, reps=300):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
def it_fused():
    p.prepare(); p.gram_fused(ds, gram)
def it_sum():
    p.prepare(); p.gram_fused_sum(ds, gram, gsum)
def it_old():
    p.prepare(); p.gram_fused(ds, gram); p.gram_sum(ds, gram, gsum2)
tag = "%s n=%d lib=%s mfma=%s" % (model, n, os.environ.get("AB_LIB", "default"), os.environ.get("VG_GRAM_FORCE_MFMA"))
print(tag, "gram_fused %.2f us | gram_fused_sum %.2f us | gram_fused+gram_sum %.2f us" % (t(it_fused), t(it_sum), t(it_old)))
it_sum(); it_old(); torch.cuda.synchronize()
print(tag, "sum agreement", float(torch.linalg.norm(gsum - gsum2) / torch.linalg.norm(gsum2)))
