import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visgeom_amd import CalibrationProblem, synthetic
for model, n in (("eucm", 10000), ("mei", 10000)):
    d = synthetic.make_mono(model, n, 1)
    p = CalibrationProblem(0)
    cam = p.add_camera(model, d["init_intrinsics"]); seq = p.add_transform(False, d["init_poses"])
    p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"]); p.finalize()
    s = p.solve(max_num_iterations=100, verbose=1)
    print(s["termination"], s["num_iterations"])
