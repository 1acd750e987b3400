"""Where a wave of the fused Gram kernel spends its life: shader-clock stamps written by lane 0 of every wave
(measurement build: python -m visgeom_amd._build --variant stamps -DVG_GRAM_STAMPS).

  AB_LIB=visgeom_amd/lib/variants/libvisgeom_amd_stamps.so python tools/exp/gram_stamps_probe.py [model] [images]

stamps: 0 entry | 1 first loads requested | 2 frame walked (LDS fence) | 3 rows of the chunk evaluated | 4 products + tree done
        | 5 per-image blocks stored, partial handed to LDS | 6 workgroup barrier passed | 7 workgroup partial stored
"""
import os
import sys

import numpy as np
import torch

root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
from visgeom_amd import _build  # noqa: E402

if os.environ.get("AB_LIB"):
    _build.LIB = os.path.join(root, os.environ["AB_LIB"])
    _build.up_to_date = lambda: True
    _build.build = lambda force=False, verbose=False: _build.LIB
from visgeom_amd import CalibrationProblem, capi, synthetic  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "eucm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
d = synthetic.make_mono(model, n, 1)
p = CalibrationProblem(0)
cam = p.add_camera(model, d["init_intrinsics"])
seq = p.add_transform(False, d["init_poses"])
ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"])
p.finalize()
gram, gsum = p.alloc_gram(ds)
n_wg = (n + 7) // 8
stamps = torch.zeros((n_wg * 4, 10), dtype=torch.int64, device="cuda")
for _ in range(20):
    p.prepare()
    p.gram_fused_sum(ds, gram, gsum)
torch.cuda.synchronize()
capi.debug_set("gram_stamps", stamps.data_ptr())
p.prepare()
p.gram_fused_sum(ds, gram, gsum)
torch.cuda.synchronize()
capi.debug_set("gram_stamps", 0)
s = stamps.cpu().numpy().astype(np.float64)
t0 = s[:, 0].min()   # NOTE: the shader clocks of different XCDs have different bases: only per-wave differences mean anything
names = ["entry->loads", "loads->walk done", "walk->rows evaluated", "products+tree", "stores", "barrier", "partial store"]
print("%s, %d images: %d waves (per-wave clock differences; the clock bases of the XCDs differ)" % (model, n, s.shape[0]))
life = s[:, 7] - s[:, 0]
print("wave life: mean %.0f  p10 %.0f  p50 %.0f  p90 %.0f cycles" % (life.mean(), *np.percentile(life, [10, 50, 90])))
for i, nm in enumerate(names):
    dt = s[:, i + 1] - s[:, i]
    print("  %-22s mean %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f" % (nm, dt.mean(), *np.percentile(dt, [10, 50, 90])))
# the launch's timeline on the 100 MHz wall clock (common to all XCDs; 10 ns ticks): when waves start and end
w0, w1 = s[:, 8], s[:, 9]
t00 = w0.min()
print("wall clock: first wave starts at 0, last wave starts at %.2f us, last wave ends at %.2f us" % ((w0.max() - t00) / 100., (w1.max() - t00) / 100.))
edges = np.arange(0., (w1.max() - t00) / 100. + 1., 1.0)
alive = [int(np.sum(((w0 - t00) / 100. <= e) & ((w1 - t00) / 100. > e))) for e in edges]
print("waves alive at every microsecond (2048 = two per SIMD):", alive)
started = [int(np.sum((w0 - t00) / 100. <= e)) for e in edges]
print("waves started by every microsecond:", started)
# the last round: waves that started after two thirds of the launch run alone on their SIMD -- their phases show what is latency
late = (w0 - t00) / 100. > 0.62 * (w1.max() - t00) / 100.
early = (w0 - t00) / 100. < 1.5
for tag, sel in (("first round (two waves per SIMD)", early), ("last round (mostly one wave per SIMD)", late)):
    if sel.sum() == 0:
        continue
    print("%s: %d waves, life mean %.0f cycles = %.2f us on the wall clock" % (tag, sel.sum(), (s[sel, 7] - s[sel, 0]).mean(), ((w1 - w0)[sel]).mean() / 100.))
    for i, nm in enumerate(names):
        dt = s[sel, i + 1] - s[sel, i]
        print("  %-22s mean %7.0f  p50 %7.0f" % (nm, dt.mean(), np.percentile(dt, 50)))
