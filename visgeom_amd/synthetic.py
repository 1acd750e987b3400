"""Deterministic synthetic calibration sets.

The reference ships no images and no corner dumps, so every benchmark /
parity workload is generated: a 12 x 8 board of 0.1 m squares seen by fisheye cameras whose
ground-truth intrinsics are the real calibrated values in data/ex_epipolar_stereo.json:2,4,6.
One portable counter-based generator (splitmix64 -> U(0,1), Box-Muller) is shared by the fixture
generator, the CPU baseline and the GPU bench: seed = 20260928 + config index, stream = image index.

Pure numpy; this module computes ground-truth projections only to place the observations -- it is
not the evaluation path (that is the HIP library) and not the oracle.
"""
import numpy as np

BASE_SEED = 20260928
IMAGE_W, IMAGE_H = 1280, 800          # data/ex_epipolar_stereo.json:17-18
BOARD_COLS, BOARD_ROWS, BOARD_SIZE = 12, 8, 0.1

GT_EUCM_CAM1 = np.array([0.595728, 0.768828, 307.318, 289.542, 642.617, 398.42])
GT_EUCM_CAM2 = np.array([0.593948, 0.774335, 307.356, 289.482, 637.871, 396.818])
GT_XI_CAM12 = np.array([0.197255, 0.000222456, -0.00421324, -0.00570702, 0.00103386, -0.0140923])
_alpha = GT_EUCM_CAM1[0]
GT_UCM = np.array([_alpha / (1 - _alpha), GT_EUCM_CAM1[2] / (1 - _alpha), GT_EUCM_CAM1[3] / (1 - _alpha),
                   GT_EUCM_CAM1[4], GT_EUCM_CAM1[5]])
GT_MEI = np.concatenate([GT_UCM[:1], [-0.05, 0.01, -0.002, 0.001, -0.0015], GT_UCM[1:]])
GT = {"eucm": GT_EUCM_CAM1, "ucm": GT_UCM, "mei": GT_MEI}
# evaluation / initial point (style of data/calib_example.json:17)
INIT = {"eucm": np.array([0.5, 1.0, 300.0, 300.0, 640.0, 400.0]),
        "ucm": np.array([1.2, 700.0, 700.0, 640.0, 400.0]),
        "mei": np.array([1.2, 0, 0, 0, 0, 0, 700.0, 700.0, 640.0, 400.0])}

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
NOISE_OFFSET = 1 << 20   # noise draws live in their own counter range of each stream
PERTURB_OFFSET = 1 << 21


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def uniform(seed, stream, counter):
    """U[0,1) number `counter` of stream `stream`: splitmix64 output k of state mix(seed, stream)."""
    with np.errstate(over="ignore"):
        stream = np.asarray(stream, dtype=np.uint64)
        counter = np.asarray(counter, dtype=np.uint64)
        s0 = _mix(np.uint64(seed) * _GOLDEN + _mix(stream + np.uint64(1)))
        z = _mix(s0 + (counter + np.uint64(1)) * _GOLDEN)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal_pair(seed, stream, counter):
    """Box-Muller: two N(0,1) from uniforms 2*counter and 2*counter+1 (offset into the noise range)."""
    c = np.asarray(counter, dtype=np.uint64) * np.uint64(2) + np.uint64(NOISE_OFFSET)
    u1 = uniform(seed, stream, c)
    u2 = uniform(seed, stream, c + np.uint64(1))
    r = np.sqrt(-2.0 * np.log(1.0 - u1))   # 1-u1 in (0,1]
    return r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)


def board_points(cols=BOARD_COLS, rows=BOARD_ROWS, size=BOARD_SIZE):
    """k = i*cols + j -> (size*j, size*i, 0)   (ordering of unified_calibration.cpp:286-292)."""
    j, i = np.meshgrid(np.arange(cols), np.arange(rows))
    return np.stack([size * j.ravel(), size * i.ravel(), np.zeros(cols * rows)], axis=1).astype(np.float64)


# ------------------------------------------------------------------ ground-truth geometry (numpy)
def rodrigues(r):
    """rotation vectors [...,3] -> matrices [...,3,3] (exact formula)."""
    r = np.asarray(r, dtype=np.float64)
    th = np.linalg.norm(r, axis=-1)[..., None, None]
    safe = np.where(th > 0, th, 1.0)
    k = r[..., None, :] / safe  # placeholder to build hat
    kx, ky, kz = (r[..., 0] / safe[..., 0, 0], r[..., 1] / safe[..., 0, 0], r[..., 2] / safe[..., 0, 0])
    z = np.zeros_like(kx)
    Kh = np.stack([np.stack([z, -kz, ky], -1), np.stack([kz, z, -kx], -1), np.stack([-ky, kx, z], -1)], -2)
    del k
    return np.eye(3) + np.sin(th) * Kh + (1 - np.cos(th)) * (Kh @ Kh)


def rotvec_from_matrix(R):
    """log map, valid for angles in (0, pi) (the generator only keeps 1e-3 < angle < pi - 0.2)."""
    tr = np.clip((np.trace(R, axis1=-2, axis2=-1) - 1) / 2, -1, 1)
    th = np.arccos(tr)
    v = np.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], -1)
    s = 2 * np.sin(th)
    s = np.where(np.abs(s) > 1e-300, s, 1.0)
    return v / s[..., None] * th[..., None]


def project(model, p, X):
    """ground-truth projection of points X[...,3]; returns (uv[...,2], ok[...])."""
    x, y, z = X[..., 0], X[..., 1], X[..., 2]
    if model == "eucm":
        a, b, fu, fv, u0, v0 = p
        d = a * np.sqrt(z * z + b * (x * x + y * y)) + (1 - a) * z
        ok = d >= 1e-3
        ds = np.where(ok, d, 1.0)
        if a > 0.5:
            ok &= (z / ds) >= (a - 1) / (2 * a - 1)
        return np.stack([fu * x / ds + u0, fv * y / ds + v0], -1), ok
    xi = p[0]
    rho = np.sqrt(x * x + y * y + z * z)
    den = z + xi * rho
    ok = den > 1e-3
    den = np.where(ok, den, 1.0)
    xn, yn = x / den, y / den
    if model == "ucm":
        return np.stack([p[1] * xn + p[3], p[2] * yn + p[4]], -1), ok
    k1, k2, k3, k4, k5, fu, fv, u0, v0 = p[1:]
    r2 = xn * xn + yn * yn
    D = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    dx = 2 * k4 * xn * yn + k5 * (r2 + 2 * xn * xn)
    dy = 2 * k5 * xn * yn + k4 * (r2 + 2 * yn * yn)
    return np.stack([fu * (xn * D + dx) + u0, fv * (yn * D + dy) + v0], -1), ok


def _draw_pose(seed, stream, attempt, board_centre):
    """board -> camera pose for each stream at attempt index `attempt` (6 uniforms per attempt)."""
    base = np.asarray(attempt, dtype=np.uint64) * np.uint64(8)
    u = [uniform(seed, stream, base + np.uint64(k)) for k in range(6)]
    rng_, off, az = 0.5 + u[0], np.deg2rad(60.0) * u[1], 2 * np.pi * u[2]
    inplane, mag, axis_az = 2 * np.pi * u[3], np.deg2rad(35.0) * u[4], 2 * np.pi * u[5]
    c = rng_[:, None] * np.stack([np.sin(off) * np.cos(az), np.sin(off) * np.sin(az), np.cos(off)], -1)
    d = c / np.linalg.norm(c, axis=-1, keepdims=True)          # board z axis = viewing ray
    helper = np.where(np.abs(d[:, :1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    e1 = np.cross(helper, d)
    e1 /= np.linalg.norm(e1, axis=-1, keepdims=True)
    e2 = np.cross(d, e1)
    R0 = np.stack([e1, e2, d], axis=-1)                        # columns
    Rz = rodrigues(np.stack([np.zeros_like(inplane), np.zeros_like(inplane), inplane], -1))
    tilt = rodrigues(mag[:, None] * np.stack([np.cos(axis_az), np.sin(axis_az), np.zeros_like(axis_az)], -1))
    R = R0 @ Rz @ tilt
    t = c - np.einsum("nij,j->ni", R, board_centre)
    return R, t


def _accept(uv, ok):
    inside = ((uv[..., 0] >= 20) & (uv[..., 0] <= IMAGE_W - 20) & (uv[..., 1] >= 20) & (uv[..., 1] <= IMAGE_H - 20))
    return (ok & inside).all(axis=-1)


def make_poses(seed, n_images, cameras, board, max_attempts=200, first=0):
    """Rejection-sample n_images board poses visible (all corners, >= 20 px inside the image) in every
    camera of `cameras` = [(model, intrinsics, R_cam_from_ref, t_cam_from_ref)], with 1e-3 < |rot| < pi-0.2.
    Returns xi [n,6] = [t, rotvec] (board -> reference camera)."""
    stream = np.arange(first, first + n_images, dtype=np.uint64)
    xi = np.zeros((n_images, 6))
    todo = np.ones(n_images, dtype=bool)
    centre = np.array([board[:, 0].max() / 2, board[:, 1].max() / 2, 0.0])
    for attempt in range(max_attempts):
        idx = np.nonzero(todo)[0]
        if idx.size == 0:
            break
        R, t = _draw_pose(seed, stream[idx], np.full(idx.size, attempt), centre)
        X = np.einsum("nij,kj->nki", R, board) + t[:, None, :]
        good = np.ones(idx.size, dtype=bool)
        for model, intr, Rc, tc in cameras:
            Xc = np.einsum("ij,nkj->nki", Rc, X) + tc
            uv, ok = project(model, intr, Xc)
            good &= _accept(uv, ok)
        r = rotvec_from_matrix(R)
        th = np.linalg.norm(r, axis=-1)
        good &= (th > 1e-3) & (th < np.pi - 0.2)
        sel = idx[good]
        xi[sel, :3] = t[good]
        xi[sel, 3:] = r[good]
        todo[sel] = False
    if todo.any():
        raise RuntimeError("pose sampling did not converge for %d images" % todo.sum())
    return xi


def _noise(seed, n_images, N, sigma, first=0):
    stream = np.repeat(np.arange(first, first + n_images, dtype=np.uint64), N)
    k = np.tile(np.arange(N, dtype=np.uint64), n_images)
    a, b = normal_pair(seed, stream, k)
    return sigma * np.stack([a, b], -1).reshape(n_images, N, 2)


def _perturb(seed, n_images, cols, lo=-0.01, hi=0.01, salt=0, first=0):
    stream = np.repeat(np.arange(first, first + n_images, dtype=np.uint64), cols)
    k = np.tile(np.arange(cols, dtype=np.uint64), n_images) + np.uint64(PERTURB_OFFSET + 64 * salt)
    return (lo + (hi - lo) * uniform(seed, stream, k)).reshape(n_images, cols)


def make_mono(model, n_images, config_index, sigma=0.1, first_image=0, gt=None):
    """Configs 2 / 4: one camera, chain [xiCamBoard DIRECT].  first_image offsets the per-image streams, so a
    rank can generate images [first_image, first_image + n_images) of a larger set (multi-GPU shards)."""
    seed = BASE_SEED + config_index
    board = board_points()
    gt = GT[model] if gt is None else np.asarray(gt, dtype=np.float64)
    poses = make_poses(seed, n_images, [(model, gt, np.eye(3), np.zeros(3))], board, first=first_image)
    X = np.einsum("nij,kj->nki", rodrigues(poses[:, 3:]), board) + poses[:, None, :3]
    uv, ok = project(model, gt, X)
    assert ok.all()
    corners = uv + _noise(seed, n_images, board.shape[0], sigma, first=first_image)
    init_poses = poses + _perturb(seed, n_images, 6, first=first_image)
    return {"model": model, "board": board, "corners": corners, "gt_intrinsics": gt.copy(), "gt_poses": poses,
            "init_intrinsics": INIT[model].copy(), "init_poses": init_poses, "seed": seed}


def make_stereo(n_pairs, config_index=3, sigma=0.1):
    """Config 3: cam-1 chain [xiCamBoard D], cam-2 chain [xiCam12 I, xiCamBoard D]
    (data/calib_stereo_example.json:51-53,88-91)."""
    seed = BASE_SEED + config_index
    board = board_points()
    R12 = rodrigues(GT_XI_CAM12[3:])
    # X2 = R12^T (X1 - t12)
    cams = [("eucm", GT_EUCM_CAM1, np.eye(3), np.zeros(3)),
            ("eucm", GT_EUCM_CAM2, R12.T, -R12.T @ GT_XI_CAM12[:3])]
    poses = make_poses(seed, n_pairs, cams, board)
    X1 = np.einsum("nij,kj->nki", rodrigues(poses[:, 3:]), board) + poses[:, None, :3]
    X2 = np.einsum("ij,nkj->nki", cams[1][2], X1) + cams[1][3]
    uv1, ok1 = project("eucm", GT_EUCM_CAM1, X1)
    uv2, ok2 = project("eucm", GT_EUCM_CAM2, X2)
    assert ok1.all() and ok2.all()
    N = board.shape[0]
    n1 = _noise(seed, n_pairs, N, sigma)
    n2 = _noise(seed + 7919, n_pairs, N, sigma)
    return {"board": board, "corners1": uv1 + n1, "corners2": uv2 + n2,
            "gt_intrinsics1": GT_EUCM_CAM1.copy(), "gt_intrinsics2": GT_EUCM_CAM2.copy(),
            "gt_xi12": GT_XI_CAM12.copy(), "gt_poses": poses,
            "init_intrinsics1": INIT["eucm"].copy(), "init_intrinsics2": INIT["eucm"].copy(),
            "init_xi12": GT_XI_CAM12 + _perturb(seed, 1, 6, salt=1)[0],
            "init_poses": poses + _perturb(seed, n_pairs, 6), "seed": seed}


def make_rig(n_frames, config_index=5, sigma=0.1):
    """Config 5: a forward-facing 2 x 2 rig [UCM, EUCM, EUCM, Mei] with 0.15 m baselines, one board seen by all
    four cameras.  Global transforms xiCam1k (k = 2..4, used INVERSE, like xiCam12 of the stereo example) and one
    per-frame xiRigBoard (DIRECT): camera 1 chain [xiRigBoard D], camera k chain [xiCam1k I, xiRigBoard D]."""
    seed = BASE_SEED + config_index
    board = board_points()
    models = ["ucm", "eucm", "eucm", "mei"]
    gts = [GT_UCM.copy(), GT_EUCM_CAM1.copy(), GT_EUCM_CAM2.copy(), GT_MEI.copy()]
    xi1k = [np.array([0.15, 0.0, 0.0, 0.004, -0.006, 0.002]),     # pose of camera k in camera 1's frame
            np.array([0.0, 0.15, 0.0, -0.005, 0.003, -0.004]),
            np.array([0.15, 0.15, 0.0, 0.002, 0.005, 0.006])]
    cams = [(models[0], gts[0], np.eye(3), np.zeros(3))]
    for k in range(3):
        R = rodrigues(xi1k[k][3:])
        cams.append((models[k + 1], gts[k + 1], R.T, -R.T @ xi1k[k][:3]))   # X_k = R^T (X_1 - t)
    poses = make_poses(seed, n_frames, cams, board)
    X1 = np.einsum("nij,kj->nki", rodrigues(poses[:, 3:]), board) + poses[:, None, :3]
    corners = []
    for k, (m, intr, Rc, tc) in enumerate(cams):
        uv, ok = project(m, intr, np.einsum("ij,nkj->nki", Rc, X1) + tc)
        assert ok.all()
        corners.append(uv + _noise(seed + 101 * k, n_frames, board.shape[0], sigma))
    return {"board": board, "models": models, "corners": corners, "gt_intrinsics": gts,
            "init_intrinsics": [INIT[m].copy() for m in models], "gt_xi1k": xi1k,
            "init_xi1k": [x + _perturb(seed, 1, 6, salt=2 + k)[0] for k, x in enumerate(xi1k)],
            "gt_poses": poses, "init_poses": poses + _perturb(seed, n_frames, 6), "seed": seed}


def _se3(xi):
    T = np.eye(4)
    T[:3, :3] = rodrigues(np.asarray(xi[3:], dtype=np.float64))
    T[:3, 3] = xi[:3]
    return T


def _xi_of(T):
    return np.concatenate([T[:3, 3], rotvec_from_matrix(T[:3, :3])])


def make_handeye(n_frames, config_index=6, sigma=0.1, odo_sigma=0.002):
    """Hand-eye set for the odometry path (data type "odometry", unified_calibration.cpp:743-807): an EUCM camera
    mounted on a moving base at xiBaseCam, a fixed board at xiOdomBoard, one base pose xiOdomBase_i per frame, and
    odometry measurements of those poses.  Camera chain [xiBaseCam INVERSE, xiOdomBase INVERSE, xiOdomBoard DIRECT]:
    X_cam = T_BC^-1 T_OB_i^-1 T_OBoard X_board.  xiOdomBase_0 is the odometry origin (identity)."""
    seed = BASE_SEED + config_index
    board = board_points()
    gt = GT_EUCM_CAM1.copy()
    cam_board = make_poses(seed, n_frames, [("eucm", gt, np.eye(3), np.zeros(3))], board)   # board -> camera, per frame
    xi_bc = np.array([0.10, -0.05, 0.20, 0.03, -0.02, 0.05])
    T_bc = _se3(xi_bc)
    T_oboard = T_bc @ _se3(cam_board[0])
    base = np.stack([_xi_of(T_oboard @ np.linalg.inv(_se3(cb)) @ np.linalg.inv(T_bc)) for cb in cam_board])
    base[0] = 0.0
    X = np.einsum("nij,kj->nki", rodrigues(cam_board[:, 3:]), board) + cam_board[:, None, :3]
    uv, ok = project("eucm", gt, X)
    assert ok.all()
    odo = base + odo_sigma * np.stack(normal_pair(seed, np.repeat(np.arange(n_frames, dtype=np.uint64), 3),
                                                  np.tile(np.arange(3, dtype=np.uint64), n_frames) + np.uint64(PERTURB_OFFSET + 4096)),
                                      -1).reshape(n_frames, 6)
    odo[0] = 0.0
    return {"board": board, "corners": uv + _noise(seed, n_frames, board.shape[0], sigma), "gt_intrinsics": gt,
            "init_intrinsics": INIT["eucm"].copy(), "gt_xi_base_cam": xi_bc, "gt_xi_odom_board": _xi_of(T_oboard),
            "init_xi_base_cam": xi_bc + _perturb(seed, 1, 6, salt=9)[0],
            "init_xi_odom_board": _xi_of(T_oboard) + _perturb(seed, 1, 6, salt=10)[0],
            "gt_base": base, "odometry": odo, "seed": seed}


GT_WHEELS = np.array([0.05, 0.05, 0.30])        # [radius_left, radius_right, track_gauge], metres
INIT_WHEELS = np.array([0.052, 0.0485, 0.315])


def wheel_step(intr, dq):
    """One wheel increment [d_left, d_right] (radians) -> the base motion [v, 0, 0, 0, 0, w] of
    odometry_cost_function.cpp:10-36: v = (r1 dl + r2 dr) / 2, w = (r2 dr - r1 dl) / g."""
    r1, r2, g = intr
    return np.array([(r1 * dq[0] + r2 * dq[1]) / 2, 0, 0, 0, 0, (r2 * dq[1] - r1 * dq[0]) / g])


def make_wheeled(n_frames, steps=5, config_index=7, sigma=0.1, q_sigma=2e-3):
    """Differential-drive set for the odometry_intrinsic path (unified_calibration.cpp:660-742): an EUCM camera
    looking forward from a wheeled base that weaves in front of a fixed board; between consecutive frames the base
    integrates `steps` wheel increments [d_left, d_right] (radians).  Same camera chain as make_handeye
    [xiBaseCam INVERSE, xiOdomBase INVERSE, xiOdomBoard DIRECT]; xiOdomBase_0 = identity.  The measured increments
    carry q_sigma of noise; GT_WHEELS generated the motion, INIT_WHEELS is where a calibration would start."""
    seed = BASE_SEED + config_index
    board = board_points()
    gt = GT_EUCM_CAM1.copy()
    centre = np.array([board[:, 0].max() / 2, board[:, 1].max() / 2, 0.0])
    # camera z = base x (forward), camera x = -base y, camera y = -base z, plus a small mounting error
    R_bc = np.array([[0.0, 0, 1], [-1, 0, 0], [0, -1, 0]]) @ rodrigues(np.array([0.03, -0.02, 0.015]))
    T_bc = np.eye(4)
    T_bc[:3, :3] = R_bc
    T_bc[:3, 3] = [0.10, 0.02, 0.30]
    T_cb0 = _se3(np.array([0, 0, 0.9, 0.10, -0.15, 0.05]))
    T_cb0[:3, 3] -= T_cb0[:3, :3] @ centre
    T_oboard = T_bc @ T_cb0
    k = np.arange((n_frames - 1) * steps, dtype=np.uint64)
    zero = np.zeros_like(k)
    # slow weave: a common forward/backward term and a differential term, both smooth in time, plus jitter
    t = np.arange((n_frames - 1) * steps) / max(1, (n_frames - 1) * steps)
    common = 0.35 * np.sin(2 * np.pi * 2 * t) + 0.1 * (uniform(seed, zero, k + np.uint64(PERTURB_OFFSET + 8192)) - 0.5)
    diff = 0.18 * np.cos(2 * np.pi * 3 * t) + 0.1 * (uniform(seed, zero + np.uint64(1), k + np.uint64(PERTURB_OFFSET + 8192)) - 0.5)
    dq_gt = np.stack([common - diff, common + diff], -1).reshape(n_frames - 1, steps, 2)
    T = np.eye(4)
    base = [np.zeros(6)]
    for i in range(n_frames - 1):
        for s_ in range(steps):
            T = T @ _se3(wheel_step(GT_WHEELS, dq_gt[i, s_]))
        base.append(_xi_of(T))
    base = np.stack(base)
    cam_board = np.stack([np.linalg.inv(T_bc) @ np.linalg.inv(_se3(b)) @ T_oboard for b in base])
    X = np.einsum("nij,kj->nki", cam_board[:, :3, :3], board) + cam_board[:, None, :3, 3]
    uv, ok = project("eucm", gt, X)
    if not _accept(uv, ok).all():
        raise RuntimeError("the board leaves the image of the wheeled base")
    na, nb = normal_pair(seed, np.repeat(np.arange(n_frames - 1, dtype=np.uint64), steps),
                         np.tile(np.arange(steps, dtype=np.uint64), n_frames - 1) + np.uint64(PERTURB_OFFSET + 12288))
    dq = dq_gt + q_sigma * np.stack([na, nb], -1).reshape(n_frames - 1, steps, 2)
    xi_bc = _xi_of(T_bc)
    return {"board": board, "corners": uv + _noise(seed, n_frames, board.shape[0], sigma), "gt_intrinsics": gt,
            "init_intrinsics": INIT["eucm"].copy(), "gt_xi_base_cam": xi_bc, "gt_xi_odom_board": _xi_of(T_oboard),
            "init_xi_base_cam": xi_bc + _perturb(seed, 1, 6, salt=11)[0],
            "init_xi_odom_board": _xi_of(T_oboard) + _perturb(seed, 1, 6, salt=12)[0],
            "gt_base": base, "delta_q": dq, "gt_delta_q": dq_gt, "gt_wheels": GT_WHEELS.copy(),
            "init_wheels": INIT_WHEELS.copy(), "seed": seed}


def write_wheeled_json(directory, d, name="wheeled", err_v=0.05, err_w=0.05, lam=0.05, anchor=True, init=True):
    """A calibration file using the "odometry_intrinsic" data type (unified_calibration.cpp:660-742) for a set made
    by make_wheeled: wheel increments in <name>_wheels.json (a list of intervals, each a list of [d_left, d_right]),
    the sequence xiOdomBase initialised by chaining the increments under the prior wheel geometry ("init": true)."""
    import json
    import os

    board = d["board"]
    n = d["corners"].shape[0]
    corners_file, wheels_file = name + "_corners.json", name + "_wheels.json"
    with open(os.path.join(directory, corners_file), "w") as f:
        json.dump([[{"camera": "cam", "points": d["corners"][i].tolist()}] for i in range(n)], f)
    with open(os.path.join(directory, wheels_file), "w") as f:
        json.dump(d["delta_q"].tolist(), f)
    odo = {"type": "odometry_intrinsic", "transform": "xiOdomBase", "err_v": err_v, "err_w": err_w, "lambda": lam,
           "radius_left": float(d["init_wheels"][0]), "radius_right": float(d["init_wheels"][1]),
           "track_gauge": float(d["init_wheels"][2]), "init": bool(init), "anchor": bool(anchor), "data_file": wheels_file}
    grid = {"type": "ir_data", "camera": "cam", "parameters": [], "init": "none",
            "transform_chain": [{"name": "xiBaseCam", "direct": False}, {"name": "xiOdomBase", "direct": False},
                                {"name": "xiOdomBoard", "direct": True}],
            "image_width": IMAGE_W, "image_height": IMAGE_H, "data_file": corners_file,
            "object": {"points": board.tolist(), "corner_ul": 0, "corner_ur": BOARD_COLS - 1,
                       "corner_bl": BOARD_COLS * (BOARD_ROWS - 1), "corner_br": BOARD_COLS * BOARD_ROWS - 1}}
    root = {"transformations": [{"name": "xiBaseCam", "global": True, "prior": True, "constant": False,
                                 "value": d["init_xi_base_cam"].tolist()},
                                {"name": "xiOdomBoard", "global": True, "prior": True, "constant": False,
                                 "value": d["init_xi_odom_board"].tolist()},
                                {"name": "xiOdomBase", "global": False, "prior": False, "constant": False}],
            "cameras": [{"name": "cam", "type": "eucm", "constant": False, "value": d["init_intrinsics"].tolist()}],
            "data": [odo, grid]}
    path = os.path.join(directory, name + ".json")
    with open(path, "w") as f:
        json.dump(root, f, indent=1)
    return path


def write_calibration_json(directory, d, model, name="calib", camera="cam", sequence="xiCamBoard", prior=False,
                           init=True, flags=(), as_images=False, skip=()):
    """Config 1: write <name>.json (+ <name>_corners.json) in the reference's calibration schema (README.md:36-223,
    style of data/calib_example.json) for a mono set made by make_mono.  Corners travel as an ir_data-style data
    file: a list of frames, each a list of {camera, points}; frames in `skip` carry no entry for this camera
    (an empty corner list -> the image is skipped everywhere, unified_calibration.cpp:363,520,1198)."""
    import json
    import os

    board = d["board"]
    n = d["corners"].shape[0]
    frames = [[] if i in skip else [{"camera": camera, "points": d["corners"][i].tolist()}] for i in range(n)]
    corners_file = name + "_corners.json"
    with open(os.path.join(directory, corners_file), "w") as f:
        json.dump(frames, f)
    tf = {"name": sequence, "global": False, "prior": bool(prior), "constant": False}
    if prior:
        tf["value"] = d["init_poses"].tolist()
    data = {"camera": camera, "parameters": list(flags), "transform_chain": [{"name": sequence, "direct": True}],
            "init": sequence if (init and not prior) else "none"}
    if as_images:
        data.update({"type": "images", "object": {"type": "checkboard", "cols": BOARD_COLS, "rows": BOARD_ROWS,
                                                   "size": BOARD_SIZE}, "corners_file": corners_file})
    else:
        data.update({"type": "ir_data", "image_width": IMAGE_W, "image_height": IMAGE_H, "data_file": corners_file,
                     "object": {"points": board.tolist(), "corner_ul": 0, "corner_ur": BOARD_COLS - 1,
                                "corner_bl": BOARD_COLS * (BOARD_ROWS - 1), "corner_br": BOARD_COLS * BOARD_ROWS - 1}})
    root = {"transformations": [tf],
            "cameras": [{"name": camera, "type": model, "constant": False, "value": d["init_intrinsics"].tolist()}],
            "data": [data]}
    path = os.path.join(directory, name + ".json")
    with open(path, "w") as f:
        json.dump(root, f, indent=1)
    return path


def _dump_corner_frames(path, cameras_and_corners):
    """the ir_data "data_file" layout (unified_calibration.cpp:252-277): a list of frames, each a list of {camera, points};
    cameras_and_corners = [(camera name, corners [n, N, 2])], all with the same n.  Written row by row ('%.17g' round-trips a
    double) -- json.dump of 10 000 x 96 nested lists takes four seconds, this takes one."""
    n = cameras_and_corners[0][1].shape[0]
    per_cam = []
    for name, arr in cameras_and_corners:
        flat = np.ascontiguousarray(arr, dtype=np.float64).reshape(n, -1, 2)
        rows = ["[%s, %s]" % (repr(float(u)), repr(float(v))) for u, v in flat.reshape(-1, 2)]
        N = flat.shape[1]
        per_cam.append(['{"camera": "%s", "points": [%s]}' % (name, ", ".join(rows[i * N:(i + 1) * N])) for i in range(n)])
    with open(path, "w") as f:
        f.write("[")
        for i in range(n):
            f.write(("" if i == 0 else ", ") + "[" + ", ".join(pc[i] for pc in per_cam) + "]")
        f.write("]")


def write_stereo_json(directory, d, name="stereo"):
    """Config 3 as a calibration file in the shape of data/calib_stereo_example.json: two EUCM cameras, the pair's pose sequence
    initialised from scratch through camera 1 ("init": "xiCamBoard"), the global xiCam12 WITHOUT a prior, initialised through
    camera 2's dataset ("init": "xiCam12": the 4-corner pose of the first frame, then initGlobalTransform over all frames,
    unified_calibration.cpp:358-429), chain [xiCam12 inverse, xiCamBoard direct] (:51-53,88-91).  One corner file holds both
    cameras' entries of every frame."""
    import json
    import os

    corners_file = name + "_corners.json"
    _dump_corner_frames(os.path.join(directory, corners_file), [("camera1", d["corners1"]), ("camera2", d["corners2"])])
    obj = {"points": d["board"].tolist(), "corner_ul": 0, "corner_ur": BOARD_COLS - 1,
           "corner_bl": BOARD_COLS * (BOARD_ROWS - 1), "corner_br": BOARD_COLS * BOARD_ROWS - 1}

    def entry(cam, init, chain):
        return {"type": "ir_data", "camera": cam, "init": init, "parameters": [], "object": obj, "image_width": IMAGE_W,
                "image_height": IMAGE_H, "transform_chain": [{"name": n, "direct": dr} for n, dr in chain], "data_file": corners_file}

    root = {"transformations": [{"name": "xiCamBoard", "global": False, "constant": False, "prior": False},
                                {"name": "xiCam12", "global": True, "constant": False, "prior": False}],
            "cameras": [{"name": "camera1", "type": "eucm", "constant": False, "value": d["init_intrinsics1"].tolist()},
                        {"name": "camera2", "type": "eucm", "constant": False, "value": d["init_intrinsics2"].tolist()}],
            "data": [entry("camera1", "xiCamBoard", [("xiCamBoard", True)]),
                     entry("camera2", "xiCam12", [("xiCam12", False), ("xiCamBoard", True)])]}
    path = os.path.join(directory, name + ".json")
    with open(path, "w") as f:
        json.dump(root, f, indent=1)
    return path


def write_handeye_json(directory, d, name="handeye", err_v=0.05, err_w=0.05, lam=0.05, anchor=True, odometry_first=True):
    """A calibration file using the "odometry" data type (README.md odometry section, parse at
    unified_calibration.cpp:743-807) for a set made by make_handeye: the sequence xiOdomBase is initialised from
    the odometry values ("init": true) and anchored at element 0; the camera data has nothing left to initialise."""
    import json
    import os

    board = d["board"]
    n = d["corners"].shape[0]
    corners_file = name + "_corners.json"
    with open(os.path.join(directory, corners_file), "w") as f:
        json.dump([[{"camera": "cam", "points": d["corners"][i].tolist()}] for i in range(n)], f)
    odo = {"type": "odometry", "transform": "xiOdomBase", "err_v": err_v, "err_w": err_w, "lambda": lam,
           "init": True, "anchor": bool(anchor), "value": d["odometry"].tolist()}
    grid = {"type": "ir_data", "camera": "cam", "parameters": [], "init": "none",
            "transform_chain": [{"name": "xiBaseCam", "direct": False}, {"name": "xiOdomBase", "direct": False},
                                {"name": "xiOdomBoard", "direct": True}],
            "image_width": IMAGE_W, "image_height": IMAGE_H, "data_file": corners_file,
            "object": {"points": board.tolist(), "corner_ul": 0, "corner_ur": BOARD_COLS - 1,
                       "corner_bl": BOARD_COLS * (BOARD_ROWS - 1), "corner_br": BOARD_COLS * BOARD_ROWS - 1}}
    root = {"transformations": [{"name": "xiBaseCam", "global": True, "prior": True, "constant": False,
                                 "value": d["init_xi_base_cam"].tolist()},
                                {"name": "xiOdomBoard", "global": True, "prior": True, "constant": False,
                                 "value": d["init_xi_odom_board"].tolist()},
                                {"name": "xiOdomBase", "global": False, "prior": False, "constant": False}],
            "cameras": [{"name": "cam", "type": "eucm", "constant": False, "value": d["init_intrinsics"].tolist()}],
            "data": [odo, grid] if odometry_first else [grid, odo]}
    path = os.path.join(directory, name + ".json")
    with open(path, "w") as f:
        json.dump(root, f, indent=1)
    return path
