"""Before / after of the front end's host phases: the ROUND-4 library (tools/exp/r04_lib/libvisgeom_amd_r04.so, built from commit
92f0fa9; it has no phase clock) driven through the same three calls under a wall clock -- add_file (parse + pose initialisation),
compute, write_residuals -- on the file tools/bench_calib.py generates for the headline size.
    python tools/exp/calib_r04_baseline.py [images]"""
import ctypes
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visgeom_amd import benchlib  # noqa: E402

images = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
old = os.path.join(os.path.dirname(os.path.abspath(__file__)), "r04_lib", "libvisgeom_amd_r04.so")
L = ctypes.CDLL(old)
L.vg_calibration_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
L.vg_calibration_add_file.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
L.vg_calibration_compute.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
L.vg_calibration_write_residuals.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]
L.vg_last_error.restype = ctypes.c_char_p
with tempfile.TemporaryDirectory() as d:
    path, info = benchlib.write_calib_workload(d, "mono_eucm", images)
    for run in range(2):
        h = ctypes.c_void_p()
        assert L.vg_calibration_create(ctypes.byref(h), 0) == 0
        t0 = time.perf_counter()
        assert L.vg_calibration_add_file(h, path.encode()) == 0, L.vg_last_error()
        t1 = time.perf_counter()
        assert L.vg_calibration_compute(h, None, None) == 0, L.vg_last_error()
        t2 = time.perf_counter()
        assert L.vg_calibration_write_residuals(h, 0, os.path.join(d, "image_error_0.txt").encode(), None, None) == 0
        t3 = time.perf_counter()
        print("round-4 library, %d images, run %d: add_file %.3f s, compute %.3f s, write_residuals %.3f s, total %.3f s"
              % (images, run, t1 - t0, t2 - t1, t3 - t2, t3 - t0), flush=True)
