import os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from visgeom_amd import synthetic as S
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set
from visgeom_amd.calibration import GenericCameraCalibration
n = int(sys.argv[1])
d = S.make_mono("eucm", n, 0, sigma=0.1)
tmp = tempfile.mkdtemp()
t0 = time.time(); path = S.write_calibration_json(tmp, d, "eucm", prior=False, init=True); t1 = time.time()
c = GenericCameraCalibration()
c.addResiduals(path); t2 = time.time()
c.compute(max_num_iterations=200); t3 = time.time()
print("images %d: write json %.2f s, addResiduals (parse + pose init) %.2f s, compute (solve + report) %.2f s, iterations %d, intr err %.2e" % (
    n, t1 - t0, t2 - t1, t3 - t2, c.summary["num_iterations"], np.max(np.abs(c.intrinsics("cam") - d["gt_intrinsics"]) / np.maximum(np.abs(d["gt_intrinsics"]), 1))))
