"""Host-side mirrors of the reference's localization reprojection costs, on top of the C ABI (include/visgeom_amd.h,
section 6).  SURVEY 8(f) rank 5.

  MonoReprojectCost     <->  struct MonoReprojectCost     include/localization/local_cost_functions.h:159-180
  SparseReprojectCost   <->  struct SparseReprojectCost   include/localization/local_cost_functions.h:183-208
  camera_jacobian       <->  class CameraJacobian         include/projection/jacobian.h:51-119
  ReprojectSet               many blocks resident in HBM, evaluated per launch (the RANSAC hypotheses of
                             src/localization/sparse_odom.cpp:511-606)

torch only owns device memory and names the stream; every number comes from the HIP kernels (csrc/vg_local.hpp).
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


def _model(m):
    return capi.MODELS[m] if isinstance(m, str) else int(m)


class ReprojectSet:
    """vg_reproject_set: the constructor arguments of many blocks, resident on the device.

    sparse=True : blocks[b] = (x1 [n, 3], x2 [n, 3], p2 [n, 2], size [n])   SparseReprojectCost
    sparse=False: blocks[b] = (x1 [5, 3], p2 [5, 2])                         MonoReprojectCost
    """

    def __init__(self, model, intrinsics, xi_base_cam, blocks, sparse, device=0, stream=None):
        import torch

        self._torch = torch
        self._lib = capi.load()
        self.sparse, self.device = bool(sparse), device
        if stream is None and torch.cuda.is_available():
            stream = torch.cuda.current_stream(device).cuda_stream
        self.stream = stream or 0
        # a torch view of the set's stream, for the event edges of evaluate() (the NULL stream is torch's default stream)
        self._torch_stream = None
        if torch.cuda.is_available():
            self._torch_stream = (torch.cuda.ExternalStream(self.stream, device=device) if self.stream
                                  else torch.cuda.default_stream(device))
        intr, xb = _c(intrinsics), _c(xi_base_cam)
        if intr.size != capi.NUM_INTRINSICS[_model(model)] or xb.size != 6:
            raise ValueError("wrong number of intrinsics / xi_base_cam needs 6 values")
        h = ctypes.c_void_p()
        n = len(blocks)
        if self.sparse:
            counts = [np.asarray(b[0]).reshape(-1, 3).shape[0] for b in blocks]
            off = np.zeros(n + 1, dtype=np.int64)
            off[1:] = np.cumsum(counts)
            cat = lambda k, w: _c(np.concatenate([np.asarray(b[k], float).reshape(-1, w) for b in blocks]) if n else np.zeros((0, w)))
            x1, x2, p2, sz = cat(0, 3), cat(1, 3), cat(2, 2), cat(3, 1)
            if not (x1.shape[0] == x2.shape[0] == p2.shape[0] == sz.shape[0]):
                raise ValueError("x1, x2, p2 and size of a block must have the same number of points")   # asserts, .h:191-192
            capi.check(self._lib.vg_sparse_reproject_create(ctypes.byref(h), device, ctypes.c_void_p(self.stream), _model(model), _ptr(intr),
                                                            _ptr(xb), n, off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _ptr(x1), _ptr(x2),
                                                            _ptr(p2), _ptr(sz)))
            self.offsets = off
        else:
            x1 = _c(np.stack([np.asarray(b[0], float).reshape(5, 3) for b in blocks]) if n else np.zeros((0, 5, 3)))
            p2 = _c(np.stack([np.asarray(b[1], float).reshape(5, 2) for b in blocks]) if n else np.zeros((0, 5, 2)))
            capi.check(self._lib.vg_mono_reproject_create(ctypes.byref(h), device, ctypes.c_void_p(self.stream), _model(model), _ptr(intr),
                                                          _ptr(xb), n, _ptr(x1), _ptr(p2)))
            self.offsets = 5 * np.arange(n + 1, dtype=np.int64)
        self._h = h
        self.n_blocks, self.n_points = n, int(self.offsets[-1])

    def evaluate(self, xi_odom, lengths=None, want_jac=True):
        """every block in one pass; xi_odom [n_blocks, 6] (, lengths [n_blocks, 5]) as torch CUDA tensors or arrays.
        -> sparse: (residuals [total, 2], jac [total, 2, 6]);  mono: (residuals [n, 10], jac_odom [n, 10, 6], jac_len [n, 10, 5])"""
        torch = self._torch
        dev = torch.device("cuda", self.device)

        def dev_t(a, shape):
            t = a if hasattr(a, "data_ptr") else torch.from_numpy(_c(a)).to(dev)
            assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and tuple(t.shape) == shape, (t.shape, shape)
            return t

        xo = dev_t(xi_odom, (self.n_blocks, 6))
        vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        # The kernels run on the stream the set was created on; uploads and allocations above / below happen on torch's
        # CURRENT stream, which may be another one at call time (`with torch.cuda.stream(...)`).  Order the two: the set's
        # stream waits for what the current stream has queued (the uploads), the current stream waits for the kernels
        # before anybody can read -- or the caching allocator can reuse -- the outputs.
        cur = torch.cuda.current_stream(dev)
        mine = self._torch_stream if cur.cuda_stream != self.stream else None

        def enter():
            if mine is not None:
                mine.wait_stream(cur)

        def leave(*tensors):
            if mine is not None:
                cur.wait_stream(mine)
                for t in tensors:
                    if t is not None:
                        t.record_stream(mine)

        if self.sparse:
            res = torch.empty((self.n_points, 2), dtype=torch.float64, device=dev)
            jac = torch.empty((self.n_points, 2, 6), dtype=torch.float64, device=dev) if want_jac else None
            enter()
            capi.check(self._lib.vg_sparse_reproject_evaluate(self._h, vp(xo), vp(res), vp(jac)))
            leave(xo, res, jac)
            self._keep = (xo,)
            return res, jac
        ln = dev_t(lengths, (self.n_blocks, 5))
        res = torch.empty((self.n_blocks, 10), dtype=torch.float64, device=dev)
        j0 = torch.empty((self.n_blocks, 10, 6), dtype=torch.float64, device=dev) if want_jac else None
        j1 = torch.empty((self.n_blocks, 10, 5), dtype=torch.float64, device=dev) if want_jac else None
        enter()
        capi.check(self._lib.vg_mono_reproject_evaluate(self._h, vp(xo), vp(ln), vp(res), vp(j0), vp(j1)))
        leave(xo, ln, res, j0, j1)
        self._keep = (xo, ln)
        return res, j0, j1

    def evaluate_block(self, block, params, want_jacobians=True, jac_mask=None):
        """Evaluate of one block with Ceres' contract: params = [xiOdom] (sparse) / [xiOdom, lengths] (mono) ->
        (residual, [Jacobians or None])"""
        ps = [_c(p) for p in params]
        sizes = [6] if self.sparse else [6, 5]
        if len(ps) != len(sizes) or any(p.size != s for p, s in zip(ps, sizes)):
            raise ValueError("parameter blocks must have sizes %s" % sizes)
        n = int(self.offsets[block + 1] - self.offsets[block])
        res = np.empty(2 * n)
        pp = (_dp * len(ps))(*[_ptr(p) for p in ps])
        jacs, jp = None, None
        if want_jacobians:
            jac_mask = jac_mask or [True] * len(sizes)
            jacs = [np.full((2 * n, s), np.nan) if m else None for s, m in zip(sizes, jac_mask)]
            jp = (_dp * len(sizes))(*[_ptr(j) if j is not None else _dp() for j in jacs])
        fn = self._lib.vg_sparse_reproject_block_evaluate if self.sparse else self._lib.vg_mono_reproject_block_evaluate
        capi.check(fn(self._h, block, pp, _ptr(res), jp))
        return res, jacs

    def synchronize(self):
        capi.check(self._lib.vg_reproject_synchronize(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vg_reproject_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SparseReprojectCost:
    """One block, the reference's constructor and Evaluate (local_cost_functions.h:185-200)."""

    def __init__(self, camera_model, intrinsics, xVec1, xVec2, pVec2, sizeVec, xiBaseCam, device=0):
        self._set = ReprojectSet(camera_model, intrinsics, xiBaseCam, [(xVec1, xVec2, pVec2, sizeVec)], True, device)
        self.n = self._set.n_points

    def num_residuals(self):
        return 2 * self.n               # set_num_residuals(_pVec2.size() * 2), .h:193

    def parameter_block_sizes(self):
        return [6]                      # .h:194

    def Evaluate(self, params, want_jacobians=True):
        return self._set.evaluate_block(0, params, want_jacobians)

    def close(self):
        self._set.close()


class MonoReprojectCost:
    """One block: SizedCostFunction<10, 6, 5> (local_cost_functions.h:159-173)."""

    def __init__(self, camera_model, intrinsics, xVec1, pVec2, xiBaseCam, device=0):
        x1, p2 = np.asarray(xVec1, float).reshape(-1, 3), np.asarray(pVec2, float).reshape(-1, 2)
        if x1.shape[0] != 5 or p2.shape[0] != 5:
            raise ValueError("MonoReprojectCost takes exactly five points")   # asserts, .h:166-167
        self._set = ReprojectSet(camera_model, intrinsics, xiBaseCam, [(x1, p2)], False, device)

    def num_residuals(self):
        return 10

    def parameter_block_sizes(self):
        return [6, 5]

    def Evaluate(self, params, want_jacobians=True, jac_mask=None):
        return self._set.evaluate_block(0, params, want_jacobians, jac_mask)

    def close(self):
        self._set.close()


def camera_jacobian(model, intrinsics, T12, T23, X2, grad=None, device=0, want_dpdxi=True):
    """CameraJacobian(camera, T12[, T23]).dpdxi / .dfdxi for points X2 [n, 3] (torch CUDA tensor or array)
    -> (dpdxi [n, 2, 6] or None, dfdxi [n, 6] or None)"""
    import torch

    L = capi.load()
    dev = torch.device("cuda", device)
    X = X2 if hasattr(X2, "data_ptr") else torch.from_numpy(_c(X2).reshape(-1, 3)).to(dev)
    g = None
    if grad is not None:
        g = grad if hasattr(grad, "data_ptr") else torch.from_numpy(_c(grad).reshape(-1, 2)).to(dev)
    n = X.shape[0]
    dp = torch.empty((n, 2, 6), dtype=torch.float64, device=dev) if want_dpdxi else None
    df = torch.empty((n, 6), dtype=torch.float64, device=dev) if g is not None else None
    intr, a = _c(intrinsics), _c(T12)
    b = _c(T23) if T23 is not None else None
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    capi.check(L.vg_camera_jacobian_evaluate(device, ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream), _model(model), _ptr(intr),
                                             _ptr(a), _ptr(b) if b is not None else None, n, vp(X), vp(g), vp(dp), vp(df)))
    torch.cuda.current_stream(device).synchronize()
    return dp, df
