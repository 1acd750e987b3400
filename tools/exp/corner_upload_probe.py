"""corner_upload_s / refine_total_s of consecutive front-end runs in one process (first run pays the library's pinned staging)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visgeom_amd import benchlib
from visgeom_amd.calibration import GenericCameraCalibration

d = tempfile.mkdtemp()
path, info = benchlib.write_calib_workload(d, sys.argv[1] if len(sys.argv) > 1 else "mono_eucm", int(sys.argv[2]) if len(sys.argv) > 2 else 10000)
for run in range(5):
    c = GenericCameraCalibration(0)
    t0 = time.perf_counter()
    c.addResiduals(path)
    c.compute()
    total = time.perf_counter() - t0
    t = c.timings()
    print("run %d: total %.4f  corner_upload %.5f (%d blocks, %.1f MB)  refine_total %.5f (kernel %.5f)  assemble %.5f solve %.5f geometric %.5f parse %.5f" % (
        run, total, t["corner_upload_s"], t["corner_uploads"], t["corner_upload_bytes"] / 1e6, t["refine_total_s"], t["refine_kernel_s"], t["assemble_s"], t["solve_s"],
        t["geometric_init_s"], t["parse_json_s"]))
    c.close()
