"""GenericCameraCalibration -- Python mirror of the reference's front-end class
(include/calibration/unified_calibration.h:91-180) on top of the C ABI; same method names as the C++ class:
addResiduals(json_file) then compute()."""
import ctypes

import numpy as np

from . import capi


class GenericCameraCalibration:
    def __init__(self, device=0):
        self._lib = capi.load()
        h = ctypes.c_void_p()
        capi.check(self._lib.vg_calibration_create(ctypes.byref(h), device))
        self._h = h
        self.summary = None

    def addResiduals(self, info_file_name):
        """read a calibration JSON (README.md:36-223) and add its residual blocks (unified_calibration.cpp:350-356)"""
        capi.check(self._lib.vg_calibration_add_file(self._h, str(info_file_name).encode()))
        return True

    def compute(self, **options):
        """solve and return the report the reference prints (unified_calibration.cpp:39-89)"""
        opt = capi.SolveOptions()
        self._lib.vg_solve_options_init(ctypes.byref(opt))
        for k, v in options.items():
            setattr(opt, k, v)
        s = capi.SolveSummary()
        capi.check(self._lib.vg_calibration_compute(self._h, ctypes.byref(opt), ctypes.byref(s)))
        self.summary = {name: getattr(s, name) for name, _ in capi.SolveSummary._fields_}
        self.summary["message"] = s.message.decode("utf-8", "replace")
        self.summary["termination"] = capi.TERMINATION.get(s.termination, str(s.termination))
        return self.report()

    def _text(self, fn):
        n = fn(self._h, None, 0)
        buf = ctypes.create_string_buffer(int(n))
        fn(self._h, buf, n)
        return buf.value.decode("utf-8", "replace")

    def report(self):
        return self._text(self._lib.vg_calibration_report)

    def log(self):
        return self._text(self._lib.vg_calibration_log)

    def corners(self, dataset):
        """the corner lists of a dataset as parsed: a list with one [N, 2] array (or None: no corners) per image"""
        out = []
        for i in range(self._lib.vg_calibration_num_images(self._h, dataset)):
            n = ctypes.c_int64(0)
            capi.check(self._lib.vg_calibration_get_corners(self._h, dataset, i, None, ctypes.byref(n)))
            if n.value == 0:
                out.append(None)
                continue
            a = np.empty(n.value)
            capi.check(self._lib.vg_calibration_get_corners(self._h, dataset, i, a.ctypes.data_as(capi._dp), None))
            out.append(a.reshape(-1, 2))
        return out

    def timings(self):
        """vg_calibration_timings as a dict: where the wall-clock time of this object went, phase by phase (seconds)"""
        t = capi.CalibrationTimings()
        capi.check(self._lib.vg_calibration_get_timings(self._h, ctypes.byref(t)))
        return {name: getattr(t, name) for name, _ in capi.CalibrationTimings._fields_}

    def num_datasets(self):
        return self._lib.vg_calibration_num_datasets(self._h)

    def intrinsics(self, camera):
        n = ctypes.c_int(0)
        capi.check(self._lib.vg_calibration_get_intrinsics(self._h, camera.encode(), None, ctypes.byref(n)))
        out = np.empty(n.value)
        capi.check(self._lib.vg_calibration_get_intrinsics(self._h, camera.encode(), out.ctypes.data_as(capi._dp), None))
        return out

    def transform(self, name):
        """[6] for a global transform, [n, 6] for a sequence"""
        n = ctypes.c_int64(0)
        capi.check(self._lib.vg_calibration_get_transform(self._h, name.encode(), 0, None, ctypes.byref(n)))
        out = np.empty((n.value, 6))
        for i in range(n.value):
            capi.check(self._lib.vg_calibration_get_transform(self._h, name.encode(), i,
                                                              out[i].ctypes.data_as(capi._dp), None))
        return out

    def writeImageResidual(self, dataset, file_name, n_images=None):
        """image_error_<i>.txt (unified_calibration.cpp:1186-1292); returns (sigma per image, outlier count)"""
        sig = np.zeros(n_images) if n_images else None
        out = ctypes.c_int64(0)
        capi.check(self._lib.vg_calibration_write_residuals(self._h, dataset, str(file_name).encode(),
                                                            sig.ctypes.data_as(capi._dp) if sig is not None else None,
                                                            ctypes.byref(out)))
        return sig, out.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vg_calibration_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def transform_from_values(values):
    """transformFromData (include/json.h:36-67)"""
    v = np.ascontiguousarray(values, dtype=np.float64)
    out = np.empty(6)
    capi.check(capi.load().vg_transform_from_values(v.size, v.ctypes.data_as(capi._dp), out.ctypes.data_as(capi._dp)))
    return out


def refine_poses(model, intrinsics, board, corners, poses, device=0, kernel_seconds=None, **options):
    """estimateInitialGrid's per-image refinement (unified_calibration.cpp:1137-1155) for all images at once: n
    INDEPENDENT 6-DOF problems, one kernel launch (vg_refine_poses).  corners [n, N, 2], poses [n, 6] (start).
    options: fields of vg_solve_options; none = the reference's setting (Ceres defaults, 500 iterations,
    SoftLOneLoss(25)).  kernel_seconds: a list that receives the launch's own duration (HIP events).
    Returns (poses [n, 6], iterations [n], final_cost [n], termination [n])."""
    L = capi.load()
    m = capi.MODELS[model] if isinstance(model, str) else int(model)
    intr = np.ascontiguousarray(intrinsics, dtype=np.float64)
    board = np.ascontiguousarray(board, dtype=np.float64).reshape(-1, 3)
    N = board.shape[0]
    corners = np.ascontiguousarray(corners, dtype=np.float64).reshape(-1, 2 * N)
    n = corners.shape[0]
    out = np.array(poses, dtype=np.float64).reshape(n, 6).copy()
    it = np.zeros(n, dtype=np.int32)
    cost = np.zeros(n)
    term = np.zeros(n, dtype=np.int32)
    opt = None
    if options:
        opt = capi.SolveOptions()
        L.vg_solve_options_init(ctypes.byref(opt))
        opt.max_num_iterations, opt.function_tolerance, opt.gradient_tolerance, opt.parameter_tolerance = 500, 1e-6, 1e-10, 1e-8
        opt.soft_l1_scale = 25.0
        for k, v in options.items():
            if not hasattr(opt, k):
                raise TypeError("unknown solver option %r" % k)
            setattr(opt, k, v)
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int32)
    ks = ctypes.c_double(0.0)
    capi.check(L.vg_refine_poses_timed(device, None, m, intr.ctypes.data_as(dp), N, board.ctypes.data_as(dp), n,
                                       corners.ctypes.data_as(dp), out.ctypes.data_as(dp), ctypes.byref(opt) if opt is not None else None,
                                       it.ctypes.data_as(ip), cost.ctypes.data_as(dp), term.ctypes.data_as(ip), ctypes.byref(ks)))
    if kernel_seconds is not None:   # a one-element list: receives the duration of vg_pose_lm_kernel alone
        kernel_seconds[:] = [ks.value]
    return out, it, cost, term
