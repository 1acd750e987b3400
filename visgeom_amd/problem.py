"""Host-side mirrors of the reference interface, on top of the C ABI (include/visgeom_amd.h).

  GenericProjectionJac   <->  struct GenericProjectionJac        include/calibration/calib_cost_functions.h:27-62
  CalibrationProblem     <->  what GenericCameraCalibration assembles   src/calibration/unified_calibration.cpp:91-180,514-630

torch is used only to own device memory (output tensors) and to name the stream; every number is
computed by the HIP kernels behind the C ABI.
"""
import ctypes

import numpy as np

from . import capi

_dp = ctypes.POINTER(ctypes.c_double)


def _c(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.reshape(shape) if shape is not None else a


def _ptr(a):
    return a.ctypes.data_as(_dp)


class BlockGroup:
    """vg_block_group: blocks created against it share one resident problem; the first Evaluate at a new parameter
    point evaluates all of them in one pass, the others copy their rows out (include/visgeom_amd.h, "block groups").
    mode: "in_place" or "state_vector" (ceres::Solve's candidate points in a fixed-layout state array)."""

    MODES = {"in_place": 0, "state_vector": 1}

    def __init__(self, device=0, mode="in_place"):
        self._lib = capi.load()
        h = ctypes.c_void_p()
        capi.check(self._lib.vg_block_group_create(ctypes.byref(h), device, self.MODES[mode]))
        self._h = h
        self.device = device

    def invalidate(self):
        """the memory behind the pointers the blocks were called with is going away (after ceres::Solve returns): every
        block evaluates alone once more before the next pass (vg_block_group_invalidate)"""
        capi.check(self._lib.vg_block_group_invalidate(self._h))

    def stats(self):
        v = [ctypes.c_int64(0) for _ in range(4)]
        capi.check(self._lib.vg_block_group_stats(self._h, *[ctypes.byref(x) for x in v]))
        return dict(zip(("blocks", "batched", "served", "alone"), (x.value for x in v)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vg_block_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GenericProjectionJac:
    """One residual block = one image of one camera.

    Same constructor arguments, parameter-block sizes and Evaluate contract as the reference's
    cost function (calib_cost_functions.h:29-46, calib_cost_functions.cpp:28-117); the arithmetic runs
    on the GPU through vg_block_evaluate.
    """

    def __init__(self, proj, grid, model, transform_status_vec, device=0, group=None):
        L = capi.load()
        self._lib = L
        self.model = capi.MODELS[model] if isinstance(model, str) else int(model)
        self.status = [int(s) for s in transform_status_vec]
        grid = _c(grid).reshape(-1, 3)
        proj = _c(proj).reshape(-1, 2)
        if proj.shape[0] != grid.shape[0]:
            # the corner list must have exactly as many entries as the board
            raise ValueError("proj and grid must have the same number of points")
        self.N = grid.shape[0]
        st = (ctypes.c_int * max(len(self.status), 1))(*self.status)
        h = ctypes.c_void_p()
        if group is not None:
            capi.check(L.vg_block_create_in_group(ctypes.byref(h), group._h, self.model, len(self.status), st, self.N,
                                                  _ptr(grid), _ptr(proj)))
        else:
            capi.check(L.vg_block_create(ctypes.byref(h), device, self.model, len(self.status), st, self.N,
                                         _ptr(grid), _ptr(proj)))
        self._h = h

    def parameter_block_sizes(self):
        n = self._lib.vg_block_num_parameter_blocks(self._h)
        return [self._lib.vg_block_parameter_block_size(self._h, i) for i in range(n)]

    def num_residuals(self):
        return self._lib.vg_block_num_residuals(self._h)

    def Evaluate(self, params, want_jacobians=True, jac_mask=None):
        """params = [intrinsics, xi_0 .. xi_{L-1}] -> (residual[2N], list of row-major Jacobians or None).
        jac_mask[b] False passes a NULL pointer for block b (constant parameter block)."""
        sizes = getattr(self, "_sizes", None) or self.parameter_block_sizes()
        self._sizes = sizes
        ps = [_c(p) for p in params]  # contiguous float64 views are passed as they are: the pointers are the caller's
        # a block group reads these addresses again when another block opens the next pass: conversions of lists /
        # non-contiguous / non-float64 inputs are temporaries, so they are kept alive until this block's next call
        self._last_params = ps
        if len(ps) != len(sizes) or any(p.size != s for p, s in zip(ps, sizes)):
            raise ValueError("parameter blocks must have sizes %s" % sizes)
        pp = (_dp * len(ps))(*[_ptr(p) for p in ps])
        res = np.empty(self.num_residuals())
        jacs, jp = None, None
        if want_jacobians:
            if jac_mask is None:
                jac_mask = [True] * len(sizes)
            jacs = [np.full((res.size, s), np.nan) if m else None for s, m in zip(sizes, jac_mask)]
            jp = (_dp * len(sizes))(*[_ptr(j) if j is not None else _dp() for j in jacs])
        capi.check(self._lib.vg_block_evaluate(self._h, pp, _ptr(res), jp))
        return res, jacs

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vg_block_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CalibrationProblem:
    """Batched problem resident in HBM: cameras, global / sequence transforms, datasets."""

    def __init__(self, device=0, stream=None):
        import torch

        self._torch = torch
        L = capi.load()
        self._lib = L
        self.device = device
        if stream is None and torch.cuda.is_available():
            stream = torch.cuda.current_stream(device).cuda_stream
        self.stream = stream or 0
        h = ctypes.c_void_p()
        capi.check(L.vg_problem_create(ctypes.byref(h), device, ctypes.c_void_p(self.stream)))
        self._h = h
        self.cameras, self.transforms, self.datasets = [], [], []

    # -- assembly ---------------------------------------------------------------------------
    def add_camera(self, model, intrinsics, constant=False):
        m = capi.MODELS[model] if isinstance(model, str) else int(model)
        v = _c(intrinsics)
        if v.size != capi.NUM_INTRINSICS[m]:
            raise ValueError("wrong number of intrinsics")  # unified_calibration.cpp:153 throws
        cid = ctypes.c_int(-1)
        capi.check(self._lib.vg_problem_add_camera(self._h, m, _ptr(v), int(constant), ctypes.byref(cid)))
        self.cameras.append({"model": m, "K": v.size, "constant": constant})
        return cid.value

    def add_transform(self, is_global, values=None, count=1, constant=False):
        if is_global:
            count = 1
        v = None
        if values is not None:
            v = _c(values).reshape(-1, 6)
            count = v.shape[0]
        tid = ctypes.c_int(-1)
        capi.check(self._lib.vg_problem_add_transform(self._h, int(is_global), int(constant), int(count),
                                                      _ptr(v) if v is not None else None, ctypes.byref(tid)))
        self.transforms.append({"global": bool(is_global), "count": count, "constant": constant})
        return tid.value

    def add_dataset(self, camera, chain, board, corners, image_index=None):
        """chain = [(transform_id, status), ...] camera side first; corners [n_images, N, 2]."""
        board = _c(board).reshape(-1, 3)
        N = board.shape[0]
        corners = _c(corners).reshape(-1, 2 * N)
        n_img = corners.shape[0]
        tids = (ctypes.c_int * max(len(chain), 1))(*[c[0] for c in chain])
        st = (ctypes.c_int * max(len(chain), 1))(*[c[1] for c in chain])
        idx = None
        if image_index is not None:
            idx = np.ascontiguousarray(image_index, dtype=np.int32)
            if idx.size != n_img:
                raise ValueError("image_index must have one entry per image")
        did = ctypes.c_int(-1)
        capi.check(self._lib.vg_problem_add_dataset(
            self._h, camera, len(chain), tids, st, N, _ptr(board), n_img,
            idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) if idx is not None else None,
            _ptr(corners), ctypes.byref(did)))
        self.datasets.append({"camera": camera, "chain": list(chain), "N": N, "n_blocks": n_img,
                              "K": self.cameras[camera]["K"], "L": len(chain)})
        return did.value

    def add_transformation_prior(self, transform, stiffness):
        """TransformationPrior block (calib_cost_functions.h:79-103) pulling a global transform towards its current value"""
        v = _c(stiffness)
        if v.size != 6:
            raise ValueError("stiffness needs 6 values")
        capi.check(self._lib.vg_problem_add_transformation_prior(self._h, transform, _ptr(v)))

    def add_odometry_prior(self, transform, index, err_v, err_w, lam, xi1, xi2):
        """OdometryPrior block (calib_cost_functions.h:64-77) between elements index and index + 1 of a sequence"""
        a, b = _c(xi1), _c(xi2)
        capi.check(self._lib.vg_problem_add_odometry_prior(self._h, transform, index, err_v, err_w, lam, _ptr(a), _ptr(b)))

    def add_parameter_block(self, values, constant=False):
        """a free-standing global parameter block (e.g. the odometry intrinsics); returns its id"""
        v = _c(values)
        bid = ctypes.c_int(-1)
        capi.check(self._lib.vg_problem_add_parameter_block(self._h, v.size, _ptr(v), int(constant), ctypes.byref(bid)))
        return bid.value

    def parameter_block_offset(self, block):
        return self._lib.vg_problem_parameter_block_offset(self._h, block)

    def add_odometry_cost(self, transform, index, err_v, err_w, lam, delta_q, block):
        """OdometryCost block (odometry_cost_function.h:33-57) between elements index and index + 1 of a sequence and
        the 3-vector parameter block of the odometry intrinsics; delta_q [n, 2] wheel increments of the interval"""
        dq = _c(delta_q).reshape(-1, 2)
        capi.check(self._lib.vg_problem_add_odometry_cost(self._h, transform, index, err_v, err_w, lam, dq.shape[0], _ptr(dq), block))

    def set_pose_constant(self, transform, index):
        capi.check(self._lib.vg_problem_set_pose_constant(self._h, transform, index))

    def finalize(self):
        capi.check(self._lib.vg_problem_finalize(self._h))
        self.num_parameters = self._lib.vg_problem_num_parameters(self._h)
        return self

    # -- parameters -------------------------------------------------------------------------
    def camera_offset(self, cam):
        return self._lib.vg_problem_camera_offset(self._h, cam)

    def transform_offset(self, tid, index=0):
        return self._lib.vg_problem_transform_offset(self._h, tid, index)

    def set_parameters(self, params):
        v = _c(params)
        if v.size != self.num_parameters:
            raise ValueError("parameter vector has the wrong length")
        capi.check(self._lib.vg_problem_set_parameters(self._h, _ptr(v)))

    def get_parameters(self):
        v = np.empty(self.num_parameters)
        capi.check(self._lib.vg_problem_get_parameters(self._h, _ptr(v)))
        return v

    def parameters_device_ptr(self):
        return self._lib.vg_problem_parameters_device(self._h)

    # -- evaluation -------------------------------------------------------------------------
    def alloc_outputs(self, d, want_jac=True, jac_mask=None):
        """torch device tensors in the Ceres block layout for dataset d."""
        torch = self._torch
        ds = self.datasets[d]
        dev = torch.device("cuda", self.device)
        nb, N, K, L = ds["n_blocks"], ds["N"], ds["K"], ds["L"]
        if jac_mask is None:
            jac_mask = [True] * (L + 1)
        res = torch.empty((nb, 2 * N), dtype=torch.float64, device=dev)
        ji = torch.empty((nb, 2 * N, K), dtype=torch.float64, device=dev) if want_jac and jac_mask[0] else None
        jm = [torch.empty((nb, 2 * N, 6), dtype=torch.float64, device=dev) if want_jac and jac_mask[1 + l] else None
              for l in range(L)]
        return res, ji, jm

    def prepare(self):
        capi.check(self._lib.vg_problem_prepare(self._h))

    def force_prepared_frames(self, on=True):
        """tests / A-B measurements: every kernel reads the chain-prep launch's frames (vg_problem_force_prepared_frames)"""
        capi.check(self._lib.vg_problem_force_prepared_frames(self._h, int(bool(on))))

    def evaluate_dataset(self, d, res, jac_intr=None, jac_member=None):
        """kernel 2 on dataset d; outputs are torch tensors (or None) from alloc_outputs."""
        L = self.datasets[d]["L"]
        jm = (ctypes.c_void_p * max(L, 1))()
        for l in range(L):
            t = jac_member[l] if jac_member else None
            jm[l] = t.data_ptr() if t is not None else None
        capi.check(self._lib.vg_dataset_evaluate(self._h, d, ctypes.c_void_p(res.data_ptr()),
                                                 ctypes.c_void_p(jac_intr.data_ptr()) if jac_intr is not None else None,
                                                 jm))

    def evaluate_all(self, outputs):
        """every dataset in one pass (vg_problem_evaluate): outputs[d] = (res, jac_intr, [jac_member...]) as returned
        by alloc_outputs(d); datasets share launches (a stereo pair or a rig is one emit launch)."""
        arr = (capi.DatasetOutputs * len(self.datasets))()
        for d, (res, ji, jm) in enumerate(outputs):
            arr[d].residuals = res.data_ptr() if res is not None else None
            arr[d].jac_intr = ji.data_ptr() if ji is not None else None
            for l in range(self.datasets[d]["L"]):
                t = jm[l] if jm else None
                arr[d].jac_member[l] = t.data_ptr() if t is not None else None
        capi.check(self._lib.vg_problem_evaluate(self._h, arr))

    # -- normal equations ------------------------------------------------------------------
    def gram_width(self, d):
        return self._lib.vg_dataset_gram_width(self._h, d)

    def alloc_gram(self, d):
        torch = self._torch
        W = self.gram_width(d)
        dev = torch.device("cuda", self.device)
        return (torch.empty((self.datasets[d]["n_blocks"], W, W), dtype=torch.float64, device=dev),
                torch.empty((W, W), dtype=torch.float64, device=dev))

    def gram_fused(self, d, gram):
        """per-image Gram of [J | r], J never materialised (needs prepare() at the current parameters)."""
        capi.check(self._lib.vg_dataset_gram_fused(self._h, d, ctypes.c_void_p(gram.data_ptr())))

    def gram_fused_all(self, grams):
        """gram_fused for every dataset (grams[d]: its tensor), one merged launch where possible (vg_problem_gram_fused)."""
        arr = (ctypes.c_void_p * max(len(grams), 1))()
        for i, g in enumerate(grams):
            arr[i] = g.data_ptr() if g is not None else None
        capi.check(self._lib.vg_problem_gram_fused(self._h, arr))

    def gram_fused_sum_all(self, grams, sums):
        """gram_fused_all + the fixed-order sum of every dataset's blocks (sums[d]: [W, W] tensor), the sums of all merged
        datasets in ONE more launch (vg_problem_gram_fused_sum)."""
        n = max(len(grams), 1)
        ga, sa = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
        for i, (g, s) in enumerate(zip(grams, sums)):
            ga[i] = g.data_ptr() if g is not None else None
            sa[i] = s.data_ptr()
        capi.check(self._lib.vg_problem_gram_fused_sum(self._h, ga, sa))

    def gram_fused_sum(self, d, gram, out):
        """gram_fused + the fixed-order sum over the dataset's blocks in two launches (vg_dataset_gram_fused_sum)."""
        capi.check(self._lib.vg_dataset_gram_fused_sum(self._h, d, ctypes.c_void_p(gram.data_ptr()),
                                                       ctypes.c_void_p(out.data_ptr())))

    def gram_from_rows(self, d, res, jac_intr, jac_member, gram):
        """the same Gram matrices from the rows evaluate_dataset wrote (second pass)."""
        L = self.datasets[d]["L"]
        jm = (ctypes.c_void_p * max(L, 1))()
        for l in range(L):
            jm[l] = jac_member[l].data_ptr()
        capi.check(self._lib.vg_dataset_gram_from_rows(self._h, d, ctypes.c_void_p(res.data_ptr()),
                                                       ctypes.c_void_p(jac_intr.data_ptr()), jm,
                                                       ctypes.c_void_p(gram.data_ptr())))

    def gram_sum(self, d, gram, out):
        capi.check(self._lib.vg_dataset_gram_sum(self._h, d, ctypes.c_void_p(gram.data_ptr()),
                                                 ctypes.c_void_p(out.data_ptr())))

    def refine_poses(self, d, poses, kernel_seconds=None, **options):
        """estimateInitialGrid's per-image refinement (unified_calibration.cpp:1137-1155) on the dataset's resident corners
        (vg_dataset_refine_poses): n independent 6-DOF problems in one launch at the camera's current intrinsics.  poses [n, 6]:
        camera-frame board poses to start from.  Returns (poses, iterations, final_cost, termination)."""
        n = self.datasets[d]["n_blocks"]
        out = np.array(poses, dtype=np.float64, order="C").reshape(n, 6)   # one copy: the call works in place
        it, cost, term = np.empty(n, dtype=np.int32), np.empty(n), np.empty(n, dtype=np.int32)
        opt = None
        if options:
            opt = capi.SolveOptions()
            self._lib.vg_solve_options_init(ctypes.byref(opt))
            opt.max_num_iterations, opt.function_tolerance, opt.gradient_tolerance, opt.parameter_tolerance = 500, 1e-6, 1e-10, 1e-8
            opt.soft_l1_scale = 25.0
            for k, v in options.items():
                if not hasattr(opt, k):
                    raise TypeError("unknown solver option %r" % k)
                setattr(opt, k, v)
        ip = ctypes.POINTER(ctypes.c_int32)
        ks = ctypes.c_double(0.0)
        capi.check(self._lib.vg_dataset_refine_poses(self._h, d, _ptr(out), ctypes.byref(opt) if opt is not None else None,
                                                     it.ctypes.data_as(ip), _ptr(cost), term.ctypes.data_as(ip), ctypes.byref(ks)))
        if kernel_seconds is not None:
            kernel_seconds[:] = [ks.value]
        return out, it, cost, term

    # -- solve ------------------------------------------------------------------------------
    def solve(self, allreduce=None, comm=None, **options):
        """Levenberg-Marquardt with per-pose Schur elimination; replaces ceres::Solve at
        unified_calibration.cpp:53.  options: fields of vg_solve_options (defaults = the reference's
        Solver::Options, unified_calibration.cpp:42-52).  allreduce(np_array) sums a host buffer over ranks in
        place (multi-GPU through host buffers); comm = a visgeom_amd.distributed.Comm (multi-GPU through RCCL on the
        device buffers, the production route).  Returns the summary as a dict; the solution is in get_parameters()."""
        opt = capi.SolveOptions()
        self._lib.vg_solve_options_init(ctypes.byref(opt))
        for k, v in options.items():
            if not hasattr(opt, k):
                raise TypeError("unknown solver option %r" % k)
            setattr(opt, k, v)
        keep = None
        if allreduce is not None:
            def _cb(buf, n, _user):
                try:
                    allreduce(np.ctypeslib.as_array(buf, shape=(n,)))
                    return 0
                except Exception:  # never let an exception cross the C boundary
                    import traceback

                    traceback.print_exc()
                    return 1
            keep = capi.ALLREDUCE_FN(_cb)
            opt.allreduce = keep
        if comm is not None:
            opt.comm = comm.handle
        summ = capi.SolveSummary()
        capi.check(self._lib.vg_problem_solve(self._h, ctypes.byref(opt), ctypes.byref(summ)))
        out = {name: getattr(summ, name) for name, _ in capi.SolveSummary._fields_}
        out["message"] = summ.message.decode("utf-8", "replace")
        out["termination"] = capi.TERMINATION.get(summ.termination, str(summ.termination))
        return out

    def evaluate_dataset_to_host(self, d, res, jac_intr=None, jac_member=None):
        """kernel 2 + D2H of the Ceres-layout arrays into host tensors / arrays (pinned torch tensors recommended);
        what an EvaluationCallback hands to the per-block Evaluate copies."""
        L = self.datasets[d]["L"]

        def ptr(t):
            return t.data_ptr() if hasattr(t, "data_ptr") else t.ctypes.data

        jm = (ctypes.c_void_p * max(L, 1))()
        for l in range(L):
            t = jac_member[l] if jac_member else None
            jm[l] = ptr(t) if t is not None else None
        capi.check(self._lib.vg_dataset_evaluate_to_host(self._h, d, ctypes.c_void_p(ptr(res)),
                                                         ctypes.c_void_p(ptr(jac_intr)) if jac_intr is not None else None, jm))

    def synchronize(self):
        capi.check(self._lib.vg_problem_synchronize(self._h))

    def failed_count(self, d):
        n = ctypes.c_int64(0)
        capi.check(self._lib.vg_dataset_failed_count(self._h, d, ctypes.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vg_problem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
