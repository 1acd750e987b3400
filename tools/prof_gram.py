#!/usr/bin/env python3
"""profiling driver: runs each normal-equation kernel REPS times on the headline workload (for rocprofv3)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from visgeom_amd import CalibrationProblem, synthetic  # noqa: E402
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set

model = sys.argv[1] if len(sys.argv) > 1 else "eucm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
d = synthetic.make_mono(model, n, 1)
p = CalibrationProblem(0)
cam = p.add_camera(model, d["init_intrinsics"])
seq = p.add_transform(False, d["init_poses"])
ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"])
p.finalize()
res, ji, jm = p.alloc_outputs(ds)
gram, gsum = p.alloc_gram(ds)
p.prepare()
for _ in range(reps):
    p.evaluate_dataset(ds, res, ji, jm)
for _ in range(reps):
    p.gram_fused(ds, gram)
for _ in range(reps):
    p.gram_from_rows(ds, res, ji, jm, gram)
for _ in range(reps):
    p.gram_sum(ds, gram, gsum)
torch.cuda.synchronize()
p.close()
