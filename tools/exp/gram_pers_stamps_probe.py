"""Timeline of the persistent Gram kernel on the 100 MHz wall clock (measurement build:
python -m visgeom_amd._build --variant stamps -DVG_GRAM_STAMPS):
  AB_LIB=visgeom_amd/lib/variants/libvisgeom_amd_stamps.so python tools/exp/gram_pers_stamps_probe.py [images] [hook 2|3]"""
import os, sys
import numpy as np, torch
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
from visgeom_amd import _build
if os.environ.get("AB_LIB"):
    _build.LIB = os.path.join(root, os.environ["AB_LIB"])
    _build.up_to_date = lambda: True
    _build.build = lambda force=False, verbose=False: _build.LIB
from visgeom_amd import CalibrationProblem, capi, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
hook = int(sys.argv[2]) if len(sys.argv) > 2 else 3
waves = 8 if hook == 3 else 4
d = synthetic.make_mono("eucm", n, 1)
p = CalibrationProblem(0)
cam = p.add_camera("eucm", d["init_intrinsics"]); seq = p.add_transform(False, d["init_poses"])
ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"]); p.finalize()
gram, gsum = p.alloc_gram(ds)
capi.debug_set("gram_persistent", hook)
n_wg = 256 if hook == 3 else 512
stamps = torch.zeros((n_wg * waves, 16), dtype=torch.int64, device="cuda")
for _ in range(20):
    p.prepare(); p.gram_fused_sum(ds, gram, gsum)
torch.cuda.synchronize()
capi.debug_set("gram_stamps", stamps.data_ptr())
for _ in range(20):   # a train of launches: the stamps that remain are the LAST one's (no idle GPU in front of it)
    p.prepare(); p.gram_fused_sum(ds, gram, gsum)
torch.cuda.synchronize()
capi.debug_set("gram_stamps", 0)
s = stamps.cpu().numpy().astype(np.float64)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
us = lambda x: (x - t0) / 100.
print("eucm %d images, hook %d: %d waves stamped; all times in us since the first wave's entry" % (n, hook, s.shape[0]))
for i, nm in ((0, "entry"), (1, "walk + barrier done"), (2, "first pair's observations in LDS"), (14, "chunk barrier passed"), (15, "end")):
    v = us(s[:, i]); print("  %-34s min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (nm, v.min(), np.percentile(v, 50), np.percentile(v, 90), v.max()))
units = (s[:, 3:10] > 0).sum(axis=1)
print("  pairs per wave: " + " ".join("%d:%d" % (k, int((units == k).sum())) for k in range(int(units.max()) + 1)))
prev = s[:, 2]
for k in range(int(units.max())):
    sel = s[:, 3 + k] > 0
    dt = (s[sel, 3 + k] - (s[sel, 2] if k == 0 else s[sel, 2 + k])) / 100.
    e = us(s[sel, 3 + k])
    print("  pair %d: %5d waves, duration p10 %5.2f p50 %5.2f p90 %5.2f us; ends p50 %6.2f max %6.2f" % (k + 1, sel.sum(), np.percentile(dt, 10), np.percentile(dt, 50), np.percentile(dt, 90), np.percentile(e, 50), e.max()))
last = np.array([s[i, 2 + units[i]] for i in range(s.shape[0])])
idle = (s[:, 14] - last) / 100.
print("  wait at the chunk barrier behind a wave's last pair: p10 %.2f p50 %.2f p90 %.2f max %.2f us" % tuple(np.percentile(idle, [10, 50, 90, 100])))

# inside the first pair (stamps 10-12)
for a_, b_, nm in ((2, 10, "inputs read, next pair requested"), (10, 11, "rows, products, tree"), (11, 12, "wait for the next pair's observations"), (12, 3, "stores + totals")):
    dt = (s[:, b_] - s[:, a_]) / 100.
    print("  first pair, %-40s p10 %5.2f p50 %5.2f p90 %5.2f max %5.2f us" % (nm, *np.percentile(dt, [10, 50, 90, 100])))

raw = stamps.cpu().numpy()
raw = raw[raw[:, 0] > 0]
c1 = (raw[:, 13] & 0xffffffff).astype(np.float64); c2 = ((raw[:, 13] >> 32) & 0xffffffff).astype(np.float64)
w1 = (s[:, 11] - s[:, 10]) / 100.
print("  rows / products / tree in SHADER-clock cycles: first pair p10 %.0f p50 %.0f p90 %.0f | second pair p10 %.0f p50 %.0f p90 %.0f" % (*np.percentile(c1, [10, 50, 90]), *np.percentile(c2[c2 > 0], [10, 50, 90])))
ok = w1 > 0
print("  first pair: shader cycles per wall-clock microsecond p10 %.0f p50 %.0f p90 %.0f" % tuple(np.percentile(c1[ok] / w1[ok], [10, 50, 90])))
