"""ctypes binding of the CPU ORACLE (oracle/vg_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from visgeom_amd/ (the product path).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvg_oracle.so")

MODEL_EUCM, MODEL_UCM, MODEL_MEI = 0, 1, 2
MODELS = {"eucm": MODEL_EUCM, "ucm": MODEL_UCM, "mei": MODEL_MEI}
NUM_INTRINSICS = {MODEL_EUCM: 6, MODEL_UCM: 5, MODEL_MEI: 10}
DIRECT, INVERSE = 0, 1
DOUBLE_BIG = 1e15

_dp = ctypes.POINTER(ctypes.c_double)
_dpp = ctypes.POINTER(_dp)
_ip = ctypes.POINTER(ctypes.c_int)
_lp = ctypes.POINTER(ctypes.c_long)


def build(force=False):
    """Compile oracle/libvg_oracle.so with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "vg_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libvg_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.vgo_eval_block.restype = ctypes.c_int
        L.vgo_eval_block.argtypes = [ctypes.c_int, ctypes.c_int, _ip, ctypes.c_int, _dp, _dp,
                                     _dpp, _dp, _dpp]
        L.vgo_eval_dataset.restype = ctypes.c_long
        L.vgo_eval_dataset.argtypes = [ctypes.c_int, ctypes.c_int, _ip, ctypes.c_int, _dp,
                                       ctypes.c_long, _dp, _dp, ctypes.c_long, _lp, _lp, _lp,
                                       _dp, _dp, _dpp, ctypes.c_int]
        L.vgo_block_gram.restype = None
        L.vgo_block_gram.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dpp, _dp]
        L.vgo_block_gram_fast.restype = None
        L.vgo_block_gram_fast.argtypes = L.vgo_block_gram.argtypes
        for name in ("vgo_compose", "vgo_compose_inverse"):
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [_dp, _dp, _dp]
        for name in ("vgo_rotation_matrix", "vgo_inter_omega_rot", "vgo_quat_from_rotvec",
                     "vgo_quat_to_rotvec"):
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [_dp, _dp]
        L.vgo_project_point.restype = ctypes.c_int
        L.vgo_project_point.argtypes = [ctypes.c_int, _dp, _dp, _dp]
        L.vgo_projection_jacobian.restype = ctypes.c_int
        L.vgo_projection_jacobian.argtypes = [ctypes.c_int, _dp, _dp, _dp, _dp]
        L.vgo_intrinsic_jacobian.restype = ctypes.c_int
        L.vgo_intrinsic_jacobian.argtypes = [ctypes.c_int, _dp, _dp, _dp, _dp]
        L.vgo_max_threads.restype = ctypes.c_int
        L.vgo_dataset_gram.restype = ctypes.c_long
        L.vgo_dataset_gram.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long, _dp, _dp, _dpp, _dp, _dp,
                                       ctypes.c_int]
        L.vgo_transformation_prior.restype = None
        L.vgo_transformation_prior.argtypes = [_dp, _dp, _dp, _dp, _dp]
        L.vgo_odometry_prior_init.restype = None
        L.vgo_odometry_prior_init.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp, _dp, _dp, _dp]
        L.vgo_odometry_prior_eval.restype = None
        L.vgo_odometry_prior_eval.argtypes = [_dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.vgo_odometry_cost_init.restype = None
        L.vgo_odometry_cost_init.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, _dp, _dp, _dp, _dp]
        L.vgo_odometry_cost_eval.restype = None
        L.vgo_odometry_cost_eval.argtypes = [_dp, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.vgo_camera_jacobian_init.restype = None
        L.vgo_camera_jacobian_init.argtypes = [_dp, _dp, _dp, _dp, _dp]
        L.vgo_camera_jacobian_eval.restype = None
        L.vgo_camera_jacobian_eval.argtypes = [ctypes.c_int, _dp, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.vgo_triangulate_regular.restype = None
        L.vgo_triangulate_regular.argtypes = [_dp, _dp, ctypes.c_double, _dp, _dp, _dp, _dp, _dp, _dp]
        L.vgo_mono_reproject.restype = None
        L.vgo_mono_reproject.argtypes = [ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.vgo_sparse_reproject.restype = None
        L.vgo_sparse_reproject.argtypes = [ctypes.c_int, _dp, _dp, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.vgo_reconstruct_point.restype = ctypes.c_int
        L.vgo_reconstruct_point.argtypes = [ctypes.c_int, _dp, _dp, _dp]
        L.vgo_rotation_vector.restype = None
        L.vgo_rotation_vector.argtypes = [_dp, _dp]
        L.vgo_initial_grid_pose.restype = ctypes.c_int
        L.vgo_initial_grid_pose.argtypes = [ctypes.c_int, _dp, _dp, _dp, _dp]
        L.vgo_init_transform.restype = None
        L.vgo_init_transform.argtypes = [ctypes.c_int, _ip, ctypes.c_int, _dp, _dp, _dp]
        L.vgo_init_transform_range.restype = None
        L.vgo_init_transform_range.argtypes = [ctypes.c_int, _ip, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp]
        _lib = L
    return _lib


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(_dp)


def eval_block(model, status, grid, obs, params, want_jac=True, jac_mask=None):
    """GenericProjectionJac::Evaluate for one block.

    params = [intrinsics, xi_0, ..., xi_{L-1}];  returns (residual[2N], [J_0 .. J_L] or None).
    jac_mask[b] False -> that block's Jacobian pointer is NULL (constant parameter block).
    """
    L = len(status)
    grid = _c(grid).reshape(-1, 3)
    obs = _c(obs).reshape(-1, 2)
    N = grid.shape[0]
    assert obs.shape[0] == N
    K = NUM_INTRINSICS[model]
    ps = [_c(p) for p in params]
    assert len(ps) == L + 1 and ps[0].size == K and all(p.size == 6 for p in ps[1:])
    pp = (_dp * (L + 1))(*[_ptr(p) for p in ps])
    st = (ctypes.c_int * max(L, 1))(*status)
    res = np.empty(2 * N)
    jacs = None
    jp = None
    if want_jac:
        sizes = [K] + [6] * L
        jacs = [np.full((2 * N, s), np.nan) for s in sizes]
        if jac_mask is None:
            jac_mask = [True] * (L + 1)
        jp = (_dp * (L + 1))(*[_ptr(j) if m else _dp() for j, m in zip(jacs, jac_mask)])
        jacs = [j if m else None for j, m in zip(jacs, jac_mask)]
    ok = lib().vgo_eval_block(model, L, st, N, _ptr(grid), _ptr(obs), pp, _ptr(res), jp)
    assert ok == 1
    return res, jacs


def eval_dataset(model, status, grid, obs, param_vec, intr_offset, member_base, member_stride,
                 seq_index, want_jac=True, threads=1, out=None):
    """Batched blocks sharing camera + chain shape; Ceres block layout outputs."""
    L = len(status)
    grid = _c(grid).reshape(-1, 3)
    N = grid.shape[0]
    obs = _c(obs).reshape(-1, 2 * N)
    nb = obs.shape[0]
    K = NUM_INTRINSICS[model]
    pv = _c(param_vec)
    st = (ctypes.c_int * max(L, 1))(*status)
    mb = np.ascontiguousarray(member_base, dtype=np.int64).reshape(-1)
    ms = np.ascontiguousarray(member_stride, dtype=np.int64).reshape(-1)
    si = np.ascontiguousarray(seq_index, dtype=np.int64).reshape(-1)
    assert si.size == nb
    if out is None:
        res = np.empty((nb, 2 * N))
        ji = np.empty((nb, 2 * N, K)) if want_jac else None
        jm = [np.empty((nb, 2 * N, 6)) for _ in range(L)] if want_jac else None
    else:
        res, ji, jm = out
    jmp = (_dp * max(L, 1))(*[_ptr(j) for j in jm]) if want_jac and L else None
    n = lib().vgo_eval_dataset(model, L, st, N, _ptr(grid), nb, _ptr(obs), _ptr(pv), intr_offset,
                               mb.ctypes.data_as(_lp), ms.ctypes.data_as(_lp),
                               si.ctypes.data_as(_lp), _ptr(res),
                               _ptr(ji) if want_jac else None, jmp, threads)
    assert n == nb
    return res, ji, jm


def block_gram(residual, jac_intr, jac_members, fast=False):
    """[J|r]^T [J|r] of one block, (P+1)x(P+1), long-double accumulation unless fast."""
    residual = _c(residual)
    jac_intr = _c(jac_intr)
    jm = [_c(j) for j in jac_members]
    N = residual.size // 2
    K = jac_intr.shape[-1]
    L = len(jm)
    W = K + 6 * L + 1
    g = np.empty((W, W))
    jmp = (_dp * max(L, 1))(*[_ptr(j) for j in jm])
    f = lib().vgo_block_gram_fast if fast else lib().vgo_block_gram
    f(K, L, N, _ptr(residual), _ptr(jac_intr), jmp, _ptr(g))
    return g


def dataset_gram(res, jac_intr, jac_members, threads=1, out=None):
    """per-block Gram (plain double) of every block + their sum; CPU-baseline leg of the J^T J build."""
    nb, rows = res.shape
    K, L = jac_intr.shape[-1], len(jac_members)
    W = K + 6 * L + 1
    grams, total = out if out is not None else (np.empty((nb, W, W)), np.empty((W, W)))
    jmp = (_dp * max(L, 1))(*[_ptr(j) for j in jac_members])
    lib().vgo_dataset_gram(K, L, rows // 2, nb, _ptr(res), _ptr(jac_intr), jmp, _ptr(grams), _ptr(total), threads)
    return grams, total


def transformation_prior(stiffness, xi_prior, xi):
    """TransformationPrior::Evaluate -> (residual[6], jacobian[6,6])"""
    st, xp, x = _c(stiffness), _c(xi_prior), _c(xi)
    r, J = np.empty(6), np.empty((6, 6))
    lib().vgo_transformation_prior(_ptr(st), _ptr(xp), _ptr(x), _ptr(r), _ptr(J))
    return r, J


class OdometryPrior:
    """OdometryPrior of the calibration library (calib_cost_functions.h:64-77)"""

    def __init__(self, errV, errW, lam, xi1, xi2):
        self.zeta = np.empty(6)
        self.A = np.empty((6, 6))
        a, b = _c(xi1), _c(xi2)
        lib().vgo_odometry_prior_init(errV, errW, lam, _ptr(a), _ptr(b), _ptr(self.zeta), _ptr(self.A))

    def evaluate(self, xi1, xi2):
        a, b = _c(xi1), _c(xi2)
        r, J1, J2 = np.empty(6), np.empty((6, 6)), np.empty((6, 6))
        lib().vgo_odometry_prior_eval(_ptr(self.zeta), _ptr(self.A), _ptr(a), _ptr(b), _ptr(r), _ptr(J1), _ptr(J2))
        return r, J1, J2


class OdometryCost:
    """OdometryCost (src/calibration/odometry_cost_function.cpp), parameter blocks (xi1[6], xi2[6], intrinsics[3])"""

    def __init__(self, errV, errW, lam, delta_q, intr_prior):
        self.dq = _c(delta_q).reshape(-1, 2)
        self.zeta = np.empty(6)
        self.A = np.empty((6, 6))
        ip = _c(intr_prior)
        lib().vgo_odometry_cost_init(ctypes.c_double(errV), ctypes.c_double(errW), ctypes.c_double(lam), self.dq.shape[0],
                                     _ptr(self.dq), _ptr(ip), _ptr(self.zeta), _ptr(self.A))

    def evaluate(self, xi1, xi2, intr):
        a, b, c = _c(xi1), _c(xi2), _c(intr)
        r, J1, J2, J3 = np.empty(6), np.empty((6, 6)), np.empty((6, 6)), np.empty((6, 3))
        lib().vgo_odometry_cost_eval(_ptr(self.A), self.dq.shape[0], _ptr(self.dq), _ptr(a), _ptr(b), _ptr(c), _ptr(r),
                                     _ptr(J1), _ptr(J2), _ptr(J3))
        return r, J1, J2, J3


def compose(a, b, inverse=False):
    a, b, o = _c(a), _c(b), np.empty(6)
    (lib().vgo_compose_inverse if inverse else lib().vgo_compose)(_ptr(a), _ptr(b), _ptr(o))
    return o


def rotation_matrix(v):
    v, R = _c(v), np.empty(9)
    lib().vgo_rotation_matrix(_ptr(v), _ptr(R))
    return R.reshape(3, 3)


def inter_omega_rot(v):
    v, M = _c(v), np.empty(9)
    lib().vgo_inter_omega_rot(_ptr(v), _ptr(M))
    return M.reshape(3, 3)


def quat_from_rotvec(v):
    v, q = _c(v), np.empty(4)
    lib().vgo_quat_from_rotvec(_ptr(v), _ptr(q))
    return q


def quat_to_rotvec(q):
    q, v = _c(q), np.empty(3)
    lib().vgo_quat_to_rotvec(_ptr(q), _ptr(v))
    return v


def project_point(model, intr, X):
    intr, X, uv = _c(intr), _c(X), np.full(2, np.nan)
    ok = lib().vgo_project_point(model, _ptr(intr), _ptr(X), _ptr(uv))
    return bool(ok), uv


# ---- localization costs ----
def camera_jacobian(model, intr, T12, T23, X2, grad=None):
    """CameraJacobian (jacobian.h:51-119) for points X2 [n, 3] -> (dudxi [n, 6], dvdxi [n, 6], dfdxi [n, 6] or None)"""
    intr, T12, X2 = _c(intr), _c(T12), _c(X2).reshape(-1, 3)
    T23c = _c(T23) if T23 is not None else None
    L11, L12, L22 = np.empty(9), np.empty(9), np.empty(9)
    lib().vgo_camera_jacobian_init(_ptr(T12), _ptr(T23c) if T23c is not None else None, _ptr(L11), _ptr(L12), _ptr(L22))
    n = X2.shape[0]
    du, dv = np.empty((n, 6)), np.empty((n, 6))
    g = _c(grad).reshape(-1, 2) if grad is not None else None
    df = np.empty((n, 6)) if g is not None else None
    for i in range(n):
        lib().vgo_camera_jacobian_eval(model, _ptr(intr), int(T23 is not None), _ptr(L11), _ptr(L12), _ptr(L22), _ptr(X2[i]),
                                       _ptr(g[i]) if g is not None else None, _ptr(du[i]), _ptr(dv[i]),
                                       _ptr(df[i]) if df is not None else None)
    return du, dv, df


def triangulate_regular(xi, p, q, eps, want_jac=True):
    """Triangulator(xi, eps).computeRegular(p, q, res1, res2, jac1, jac2) for one pair -> (l1, l2, jac1 [6], jac2 [6])"""
    xi, p, q = _c(xi), _c(p), _c(q)
    R = _c(rotation_matrix(xi[3:]).ravel())
    t = _c(xi[:3])
    r1, r2 = np.empty(1), np.empty(1)
    j1, j2 = (np.empty(6), np.empty(6)) if want_jac else (None, None)
    lib().vgo_triangulate_regular(_ptr(R), _ptr(t), ctypes.c_double(eps), _ptr(p), _ptr(q), _ptr(r1), _ptr(r2),
                                  _ptr(j1) if want_jac else None, _ptr(j2) if want_jac else None)
    return float(r1[0]), float(r2[0]), j1, j2


def mono_reproject(model, intr, xi_base_cam, x1, p2, xi_odom, lengths, want_jac=True):
    """MonoReprojectCost::Evaluate -> (residual [10], jac_odom [10, 6], jac_len [10, 5])"""
    intr, xb, x1, p2, xo, ln = _c(intr), _c(xi_base_cam), _c(x1).reshape(5, 3), _c(p2).reshape(5, 2), _c(xi_odom), _c(lengths)
    r = np.empty(10)
    j0, j1 = (np.empty((10, 6)), np.empty((10, 5))) if want_jac else (None, None)
    lib().vgo_mono_reproject(model, _ptr(intr), _ptr(xb), _ptr(x1), _ptr(p2), _ptr(xo), _ptr(ln), _ptr(r),
                             _ptr(j0) if want_jac else None, _ptr(j1) if want_jac else None)
    return r, j0, j1


def sparse_reproject(model, intr, xi_base_cam, x1, x2, p2, size, xi_odom, want_jac=True):
    """SparseReprojectCost::Evaluate -> (residual [2n], jac [2n, 6])"""
    intr, xb, xo = _c(intr), _c(xi_base_cam), _c(xi_odom)
    x1, x2, p2, size = _c(x1).reshape(-1, 3), _c(x2).reshape(-1, 3), _c(p2).reshape(-1, 2), _c(size).ravel()
    n = x1.shape[0]
    r = np.empty(2 * n)
    J = np.empty((2 * n, 6)) if want_jac else None
    lib().vgo_sparse_reproject(model, _ptr(intr), _ptr(xb), n, _ptr(x1), _ptr(x2), _ptr(p2), _ptr(size), _ptr(xo), _ptr(r),
                               _ptr(J) if want_jac else None)
    return r, J


# ---- geometric pose initialisation ----
def reconstruct_point(model, intr, uv):
    intr, uv, X = _c(intr), _c(uv), np.full(3, np.nan)
    ok = lib().vgo_reconstruct_point(model, _ptr(intr), _ptr(uv), _ptr(X))
    return bool(ok), X


def rotation_vector(R):
    R, r = _c(np.asarray(R, float).ravel()), np.empty(3)
    lib().vgo_rotation_vector(_ptr(R), _ptr(r))
    return r


def initial_grid_pose(model, intr, board4, corners4):
    """estimateInitialGrid's 4-corner construction; board4 [4, 3] / corners4 [4, 2] in the order UL, UR, BL, BR"""
    intr, b, c, xi = _c(intr), _c(board4).reshape(12), _c(corners4).reshape(8), np.full(6, np.nan)
    ok = lib().vgo_initial_grid_pose(model, _ptr(intr), _ptr(b), _ptr(c), _ptr(xi))
    return bool(ok), xi


def init_transform(status, init_index, chain, xi):
    """getInitTransform: chain [n, 6] current values of the members, xi the camera-frame pose"""
    chain, xi, out = _c(chain).reshape(-1, 6), _c(xi), np.empty(6)
    st = (ctypes.c_int * max(len(status), 1))(*[int(s) for s in status])
    lib().vgo_init_transform(chain.shape[0], st, int(init_index), _ptr(chain), _ptr(xi), _ptr(out))
    return out


def init_transform_range(status, first_index, last_index, chain, xi):
    """getInitTransform for a member that occurs more than once: forward loop to its first, backward loop to its last occurrence"""
    chain, xi, out = _c(chain).reshape(-1, 6), _c(xi), np.empty(6)
    st = (ctypes.c_int * max(len(status), 1))(*[int(s) for s in status])
    lib().vgo_init_transform_range(chain.shape[0], st, int(first_index), int(last_index), _ptr(chain), _ptr(xi), _ptr(out))
    return out


def max_threads():
    return lib().vgo_max_threads()
