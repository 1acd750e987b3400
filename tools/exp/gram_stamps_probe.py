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
stamps = torch.zeros((n_wg * 4, 8), dtype=torch.int64, device="cuda")
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
t0 = s[:, 0].min()
names = ["entry->loads", "loads->walk done", "walk->rows evaluated", "products+tree", "stores", "barrier", "partial store"]
print("%s, %d images: %d waves; kernel span (first entry .. last end) %.0f cycles" % (model, n, s.shape[0], s[:, 7].max() - t0))
life = s[:, 7] - s[:, 0]
print("wave life: mean %.0f  p10 %.0f  p50 %.0f  p90 %.0f cycles" % (life.mean(), *np.percentile(life, [10, 50, 90])))
for i, nm in enumerate(names):
    dt = s[:, i + 1] - s[:, i]
    print("  %-22s mean %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f" % (nm, dt.mean(), *np.percentile(dt, [10, 50, 90])))
start = s[:, 0] - t0
print("wave start times: p10 %.0f p50 %.0f p90 %.0f max %.0f" % (*np.percentile(start, [10, 50, 90]), start.max()))
# how many waves are alive over time (occupancy profile), in 20 buckets of the kernel span
span = s[:, 7].max() - t0
edges = np.linspace(0, span, 21)
alive = [(np.sum((s[:, 0] - t0 <= e) & (s[:, 7] - t0 > e))) for e in edges[:-1]]
print("waves alive at 20 points of the span (2048 wave slots at 2 per SIMD):", alive)
