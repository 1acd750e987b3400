"""ctypes binding of include/visgeom_amd.h (the C ABI is the product boundary; this file is plumbing).

The library is HIP-only.  If it has not been built this module raises -- it never falls back to
a CPU implementation."""
import ctypes
import os

from . import _build

_dp = ctypes.POINTER(ctypes.c_double)
_dpp = ctypes.POINTER(_dp)
_ip = ctypes.POINTER(ctypes.c_int)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p
_vpp = ctypes.POINTER(ctypes.c_void_p)

MODEL_EUCM, MODEL_UCM, MODEL_MEI = 0, 1, 2
MODELS = {"eucm": MODEL_EUCM, "ucm": MODEL_UCM, "mei": MODEL_MEI}
NUM_INTRINSICS = {MODEL_EUCM: 6, MODEL_UCM: 5, MODEL_MEI: 10}
TRANSFORM_DIRECT, TRANSFORM_INVERSE = 0, 1
MAX_CHAIN = 5
DOUBLE_BIG = 1e15
OK = 0
ERR_INVALID_ARGUMENT, ERR_HIP, ERR_NO_DEVICE, ERR_STATE, ERR_ALLOC, ERR_NUMERIC = 1, 2, 3, 4, 5, 6

ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, _dp, ctypes.c_int64, ctypes.c_void_p)


class SolveOptions(ctypes.Structure):
    """struct vg_solve_options"""
    _fields_ = [("max_num_iterations", ctypes.c_int), ("function_tolerance", ctypes.c_double),
                ("gradient_tolerance", ctypes.c_double), ("parameter_tolerance", ctypes.c_double),
                ("initial_trust_region_radius", ctypes.c_double), ("max_trust_region_radius", ctypes.c_double),
                ("min_trust_region_radius", ctypes.c_double), ("min_relative_decrease", ctypes.c_double),
                ("min_lm_diagonal", ctypes.c_double), ("max_lm_diagonal", ctypes.c_double),
                ("use_bounds", ctypes.c_int), ("verbose", ctypes.c_int), ("soft_l1_scale", ctypes.c_double),
                ("allreduce", ALLREDUCE_FN), ("allreduce_user", ctypes.c_void_p), ("comm", ctypes.c_void_p)]


class DatasetOutputs(ctypes.Structure):
    """struct vg_dataset_outputs"""
    _fields_ = [("residuals", ctypes.c_void_p), ("jac_intr", ctypes.c_void_p), ("jac_member", ctypes.c_void_p * MAX_CHAIN)]


class SolveSummary(ctypes.Structure):
    """struct vg_solve_summary"""
    _fields_ = [("initial_cost", ctypes.c_double), ("final_cost", ctypes.c_double),
                ("num_iterations", ctypes.c_int), ("num_successful_steps", ctypes.c_int), ("termination", ctypes.c_int),
                ("gradient_max_norm", ctypes.c_double), ("final_radius", ctypes.c_double),
                ("total_seconds", ctypes.c_double), ("evaluate_seconds", ctypes.c_double),
                ("schur_seconds", ctypes.c_double), ("host_seconds", ctypes.c_double),
                ("num_global_columns", ctypes.c_int), ("num_pose_blocks", ctypes.c_int64),
                ("message", ctypes.c_char * 160)]


class CalibrationTimings(ctypes.Structure):
    """struct vg_calibration_timings"""
    _fields_ = [(n, ctypes.c_double) for n in ("read_files_s", "parse_json_s", "geometric_init_s", "refine_total_s", "refine_kernel_s",
                                               "global_init_s", "assemble_s", "solve_s", "readback_s", "residual_eval_s",
                                               "residual_format_s")] + \
               [(n, ctypes.c_int64) for n in ("refine_images", "refine_iterations", "refine_max_iterations", "json_bytes", "residual_lines")] + \
               [("corner_upload_s", ctypes.c_double), ("corner_uploads", ctypes.c_int64), ("corner_upload_bytes", ctypes.c_int64)]


TERMINATION = {0: "CONVERGENCE_FUNCTION", 1: "CONVERGENCE_GRADIENT", 2: "CONVERGENCE_PARAMETER", 3: "NO_CONVERGENCE",
               4: "RADIUS_TOO_SMALL", 5: "FAILURE"}

# every symbol include/visgeom_amd.h declares: (restype, argtypes)
SIGNATURES = {
    "vg_abi_version": (ctypes.c_int, []),
    "vg_last_error": (ctypes.c_char_p, []),
    "vg_device_count": (ctypes.c_int, []),
    "vg_num_intrinsics": (ctypes.c_int, [ctypes.c_int]),
    "vg_intrinsic_bounds": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _dp, _dp]),
    "vg_block_create": (ctypes.c_int, [_vpp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip, ctypes.c_int, _dp, _dp]),
    "vg_block_num_residuals": (ctypes.c_int, [_vp]),
    "vg_block_num_parameter_blocks": (ctypes.c_int, [_vp]),
    "vg_block_parameter_block_size": (ctypes.c_int, [_vp, ctypes.c_int]),
    "vg_block_evaluate": (ctypes.c_int, [_vp, _dpp, _dp, _dpp]),
    "vg_block_destroy": (None, [_vp]),
    "vg_block_group_create": (ctypes.c_int, [_vpp, ctypes.c_int, ctypes.c_int]),
    "vg_block_create_in_group": (ctypes.c_int, [_vpp, _vp, ctypes.c_int, ctypes.c_int, _ip, ctypes.c_int, _dp, _dp]),
    "vg_block_group_invalidate": (ctypes.c_int, [_vp]),
    "vg_block_group_stats": (ctypes.c_int, [_vp, _i64p, _i64p, _i64p, _i64p]),
    "vg_block_group_destroy": (None, [_vp]),
    "vg_problem_create": (ctypes.c_int, [_vpp, ctypes.c_int, _vp]),
    "vg_problem_destroy": (None, [_vp]),
    "vg_problem_add_camera": (ctypes.c_int, [_vp, ctypes.c_int, _dp, ctypes.c_int, _ip]),
    "vg_problem_add_transform": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _ip]),
    "vg_problem_add_dataset": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _ip, _ip, ctypes.c_int, _dp,
                                              ctypes.c_int64, _i32p, _dp, _ip]),
    "vg_problem_add_transformation_prior": (ctypes.c_int, [_vp, ctypes.c_int, _dp]),
    "vg_problem_add_odometry_prior": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_double,
                                                     ctypes.c_double, _dp, _dp]),
    "vg_problem_add_parameter_block": (ctypes.c_int, [_vp, ctypes.c_int, _dp, ctypes.c_int, _ip]),
    "vg_problem_parameter_block_offset": (ctypes.c_int64, [_vp, ctypes.c_int]),
    "vg_problem_add_odometry_cost": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_double,
                                                    ctypes.c_double, ctypes.c_int, _dp, ctypes.c_int]),
    "vg_odometry_cost_evaluate": (ctypes.c_int, [ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, _dp, _dp, _dp,
                                                 _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "vg_odometry_prior_evaluate": (ctypes.c_int, [ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp, _dp, _dp, _dp, _dp,
                                                  _dp, _dp]),
    "vg_problem_set_pose_constant": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64]),
    "vg_problem_finalize": (ctypes.c_int, [_vp]),
    "vg_problem_num_parameters": (ctypes.c_int64, [_vp]),
    "vg_problem_camera_offset": (ctypes.c_int64, [_vp, ctypes.c_int]),
    "vg_problem_transform_offset": (ctypes.c_int64, [_vp, ctypes.c_int, ctypes.c_int64]),
    "vg_problem_set_parameters": (ctypes.c_int, [_vp, _dp]),
    "vg_problem_get_parameters": (ctypes.c_int, [_vp, _dp]),
    "vg_problem_parameters_device": (_vp, [_vp]),
    "vg_problem_num_datasets": (ctypes.c_int, [_vp]),
    "vg_dataset_num_blocks": (ctypes.c_int64, [_vp, ctypes.c_int]),
    "vg_dataset_num_points": (ctypes.c_int, [_vp, ctypes.c_int]),
    "vg_dataset_chain_len": (ctypes.c_int, [_vp, ctypes.c_int]),
    "vg_dataset_single_launch": (ctypes.c_int, [_vp, ctypes.c_int]),
    "vg_dataset_num_intrinsics": (ctypes.c_int, [_vp, ctypes.c_int]),
    "vg_problem_prepare": (ctypes.c_int, [_vp]),
    "vg_problem_force_prepared_frames": (ctypes.c_int, [_vp, ctypes.c_int]),
    "vg_dataset_evaluate": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vpp]),
    "vg_problem_evaluate": (ctypes.c_int, [_vp, ctypes.POINTER(DatasetOutputs)]),
    "vg_problem_synchronize": (ctypes.c_int, [_vp]),
    "vg_dataset_evaluate_to_host": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vpp]),
    "vg_dataset_failed_count": (ctypes.c_int, [_vp, ctypes.c_int, _i64p]),
    "vg_dataset_gram_width": (ctypes.c_int, [_vp, ctypes.c_int]),
    "vg_dataset_gram_fused": (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    "vg_problem_gram_fused": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_void_p)]),
    "vg_problem_gram_fused_sum": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]),
    "vg_dataset_gram_fused_sum": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp]),
    "vg_dataset_gram_from_rows": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vpp, _vp]),
    "vg_dataset_gram_sum": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp]),
    "vg_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "vg_comm_create": (ctypes.c_int, [_vpp, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "vg_comm_adopt": (ctypes.c_int, [_vpp, _vp, ctypes.c_int]),
    "vg_comm_create_replicated": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int]),
    "vg_comm_create_local": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int]),
    "vg_comm_size": (ctypes.c_int, [_vp]),
    "vg_comm_rank": (ctypes.c_int, [_vp]),
    "vg_comm_allreduce_sum": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _vp]),
    "vg_comm_destroy": (None, [_vp]),
    "vg_solve_options_init": (None, [ctypes.POINTER(SolveOptions)]),
    "vg_release_cached_memory": (None, []),
    "vg_problem_solve": (ctypes.c_int, [_vp, ctypes.POINTER(SolveOptions), ctypes.POINTER(SolveSummary)]),
    "vg_refine_poses": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int, _dp, ctypes.c_int, _dp, ctypes.c_int64, _dp, _dp,
                                       ctypes.POINTER(SolveOptions), _i32p, _dp, _i32p]),
    "vg_refine_poses_timed": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int, _dp, ctypes.c_int, _dp, ctypes.c_int64, _dp, _dp,
                                             ctypes.POINTER(SolveOptions), _i32p, _dp, _i32p, _dp]),
    "vg_dataset_refine_poses": (ctypes.c_int, [_vp, ctypes.c_int, _dp, ctypes.POINTER(SolveOptions), _i32p, _dp, _i32p, _dp]),
    "vg_host_cholesky_solve": (ctypes.c_int, [ctypes.c_int, _dp, _dp, _dp]),
    "vg_calibration_create": (ctypes.c_int, [_vpp, ctypes.c_int]),
    "vg_calibration_destroy": (None, [_vp]),
    "vg_calibration_add_file": (ctypes.c_int, [_vp, ctypes.c_char_p]),
    "vg_calibration_compute": (ctypes.c_int, [_vp, ctypes.POINTER(SolveOptions), ctypes.POINTER(SolveSummary)]),
    "vg_calibration_report": (ctypes.c_int64, [_vp, ctypes.c_char_p, ctypes.c_int64]),
    "vg_calibration_log": (ctypes.c_int64, [_vp, ctypes.c_char_p, ctypes.c_int64]),
    "vg_calibration_num_datasets": (ctypes.c_int, [_vp]),
    "vg_calibration_get_intrinsics": (ctypes.c_int, [_vp, ctypes.c_char_p, _dp, _ip]),
    "vg_calibration_get_transform": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int64, _dp, _i64p]),
    "vg_calibration_write_residuals": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_char_p, _dp, _i64p]),
    "vg_calibration_get_corners": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64, _dp, _i64p]),
    "vg_calibration_num_images": (ctypes.c_int64, [_vp, ctypes.c_int]),
    "vg_calibration_get_timings": (ctypes.c_int, [_vp, ctypes.POINTER(CalibrationTimings)]),
    "vg_reconstruct_point": (ctypes.c_int, [ctypes.c_int, _dp, _dp, _dp]),
    "vg_initial_grid_pose": (ctypes.c_int, [ctypes.c_int, _dp, _dp, _dp, _dp]),
    "vg_init_transform": (ctypes.c_int, [ctypes.c_int, _ip, ctypes.c_int, _dp, _dp, _dp]),
    "vg_init_transform_range": (ctypes.c_int, [ctypes.c_int, _ip, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp]),
    "vg_transform_from_values": (ctypes.c_int, [ctypes.c_int, _dp, _dp]),
    "vg_sparse_reproject_create": (ctypes.c_int, [_vpp, ctypes.c_int, _vp, ctypes.c_int, _dp, _dp, ctypes.c_int64, _i64p, _dp, _dp, _dp, _dp]),
    "vg_mono_reproject_create": (ctypes.c_int, [_vpp, ctypes.c_int, _vp, ctypes.c_int, _dp, _dp, ctypes.c_int64, _dp, _dp]),
    "vg_reproject_num_blocks": (ctypes.c_int64, [_vp]),
    "vg_reproject_num_points": (ctypes.c_int64, [_vp]),
    "vg_reproject_block_offset": (ctypes.c_int64, [_vp, ctypes.c_int64]),
    "vg_sparse_reproject_evaluate": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "vg_mono_reproject_evaluate": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "vg_sparse_reproject_block_evaluate": (ctypes.c_int, [_vp, ctypes.c_int64, _dpp, _dp, _dpp]),
    "vg_mono_reproject_block_evaluate": (ctypes.c_int, [_vp, ctypes.c_int64, _dpp, _dp, _dpp]),
    "vg_reproject_synchronize": (ctypes.c_int, [_vp]),
    "vg_reproject_destroy": (None, [_vp]),
    "vg_camera_jacobian_evaluate": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int, _dp, _dp, _dp, ctypes.c_int64, _vp, _vp, _vp, _vp]),
    "vg_debug_set": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_longlong]),
    "vg_calib_stream_write": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_double]),
    "vg_calib_stream_copy": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64]),
    "vg_calib_d2h_copies": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, ctypes.c_int, _dp]),
    "vg_calib_fp64_fma": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]),
}


class VisgeomError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("visgeom_amd error %d: %s" % (code, msg))
        self.code = code


_lib = None
_has_hooks = False


def lib_path():
    """the library file this process loads: the default (hooks) build, or -- for the TEST HARNESS only, the library itself
    reads no environment variable -- the file VISGEOM_AMD_LIBRARY names ("production" = lib/production/libvisgeom_amd.so)"""
    sel = os.environ.get("VISGEOM_AMD_LIBRARY")
    if not sel:
        return _build.LIB
    return _build.PRODUCTION_LIB if sel == "production" else sel


class NoDebugHooks(VisgeomError):
    """vg_debug_set on a production library (built without VG_DEBUG_HOOKS: the entry is not exported)"""

    def __init__(self):
        VisgeomError.__init__(self, ERR_STATE, "this library has no debug hooks (production build)")


def has_debug_hooks():
    load()
    return _has_hooks


def load():
    """Load libvisgeom_amd.so.  Raises if it is missing -- build it with
    `python -c "import __graft_entry__ as g; g.build()"`; there is no CPU fallback."""
    global _lib
    global _has_hooks
    if _lib is not None:
        return _lib
    path = lib_path()
    if path != _build.LIB:
        if not os.path.exists(path):
            raise ImportError("VISGEOM_AMD_LIBRARY: %s does not exist (python -m visgeom_amd._build --production)" % path)
    elif not os.path.exists(_build.LIB):
        try:  # build the HIP library in-tree (hipcc cross-compiles); never substitute anything for it
            _build.build()
        except Exception as e:
            raise ImportError("%s is missing and could not be built (%s): run `python -m visgeom_amd._build` "
                              "(needs hipcc). visgeom_amd is HIP-only and has no CPU fallback." % (_build.LIB, e))
    try:  # make torch's bundled HIP runtime the one this process uses (same SONAME, libamdhip64.so.7)
        import torch  # noqa: F401
    except Exception:  # the library also works stand-alone against /opt/rocm
        pass
    L = ctypes.CDLL(path)
    _has_hooks = hasattr(L, "vg_debug_set")
    for name, (res, args) in SIGNATURES.items():
        if name == "vg_debug_set" and not _has_hooks:   # the production library: the one entry it does not export
            continue
        f = getattr(L, name)  # AttributeError here = header / library mismatch
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(code):
    if code != OK:
        raise VisgeomError(code, load().vg_last_error().decode("utf-8", "replace"))


def debug_set(name, value):
    """measurement / test hook of the library (vg_debug_set): value 0 restores the default"""
    L = load()
    if not _has_hooks:
        raise NoDebugHooks()
    check(L.vg_debug_set(name.encode(), int(value)))


def hooks_from_env():
    """tools/exp probes only: the A/B switches used to be environment variables of the library (VG_GRAM_FORCE_MFMA=1 ...);
    the library no longer reads any, this maps the old names onto vg_debug_set so that the probes still run"""
    legacy = {"VG_INLINE_CHAIN_MAX_BYTES": "inline_chain_max_bytes", "VG_GRAM_FORCE_MFMA": "gram_force_mfma", "VG_GRAM_CH1": "gram_ch1",
              "VG_GRAM_NO_MERGE": "gram_no_merge", "VG_MAX_OBS_PER_LAUNCH": "max_obs_per_launch", "VG_SOLVER_TIMING": "solver_timing",
              "VG_SOLVER_HOST_LOOP": "solver_host_loop", "VG_SOLVER_DEVICE_LOOP": "solver_device_loop",
              "VG_SOLVER_NO_SPECULATION": "solver_no_speculation", "VG_EMIT_EQUAL_TILES": "emit_equal_tiles", "VG_SCHUR_PRIVATE_GATHER": "schur_private_gather", "VG_SOLVER_EVENT_WAIT": "solver_event_wait", "VG_SOLVER_NO_FOLD_FRAMES": "solver_no_fold_frames", "VG_SOLVER_ONE_WAVE_FOLD": "solver_one_wave_fold", "VG_SOLVER_FOLD_MAX_GROUPS": "solver_fold_max_groups"}
    for env, name in legacy.items():
        if env in os.environ:
            v = os.environ[env]
            debug_set(name, int(v) if v.lstrip("-").isdigit() else 1)
