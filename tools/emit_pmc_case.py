"""One emit configuration, `launches` launches, nothing else: the target of tools/pmc_emit_mem.sh (rocprofv3 --pmc passes).
python tools/emit_pmc_case.py <images> <map window W, 0 = contiguous eighths> [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from visgeom_amd import CalibrationProblem, capi, synthetic  # noqa: E402

n, W = int(sys.argv[1]), int(sys.argv[2])
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 12
d = synthetic.make_mono("eucm", n, 1)
p = CalibrationProblem(0)
cam = p.add_camera("eucm", d["init_intrinsics"])
seq = p.add_transform(False, d["init_poses"])
ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"])
p.finalize()
p.force_prepared_frames(True)
p.prepare()
res, ji, jm = p.alloc_outputs(ds)
if capi.has_debug_hooks():
    capi.debug_set("emit_map_window", W)
for _ in range(launches):
    p.evaluate_dataset(ds, res, ji, jm)
torch.cuda.synchronize()
p.close()
