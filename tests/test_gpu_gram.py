"""Normal-equation build on the GPU vs its mathematical definition on the oracle's rows:
G_b = [J|r]^T [J|r] accumulated in long double ("JtJ/Jtr ... check GPU result vs a CPU
long-double accumulation ... <= 1e-10 relative (block-norm)")."""
import numpy as np
import pytest

from oracle import vgo

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(7)
TOL = 1e-10


@pytest.fixture(scope="module")
def vg():
    import torch

    assert torch.cuda.is_available()
    import visgeom_amd

    return visgeom_amd


def oracle_grams(model, status, board, corners, pv, intr_off, bases, strides, seq):
    r, ji, jm = vgo.eval_dataset(vgo.MODELS[model], status, board, corners, pv, intr_off, bases, strides, seq, threads=4)
    return np.stack([vgo.block_gram(r[b], ji[b], [m[b] for m in jm]) for b in range(r.shape[0])])


def assert_gram_parity(G, Gref, what=""):
    for b in range(Gref.shape[0]):
        n = np.linalg.norm(Gref[b])
        assert np.linalg.norm(G[b] - Gref[b]) <= TOL * n, "%s block %d norm" % (what, b)
        d = np.sqrt(np.abs(np.diag(Gref[b])))
        scale = np.maximum(np.outer(d, d), 1e-300)
        assert np.max(np.abs(G[b] - Gref[b]) / scale) <= TOL, "%s block %d elementwise" % (what, b)
        assert np.array_equal(G[b], G[b].T), "Gram must be exactly symmetric"


def build(vg, model, n_images, chain_kind, n_points=None):
    from visgeom_amd import synthetic as S

    d = S.make_mono(model, n_images, 2)
    board, corners = d["board"], d["corners"]
    if n_points is not None:
        board = np.concatenate([RNG.uniform(0, 1.1, (n_points, 1)), RNG.uniform(0, 0.7, (n_points, 1)),
                                RNG.uniform(-0.05, 0.05, (n_points, 1))], axis=1)
        corners = RNG.uniform(50, 1200, (n_images, n_points, 2))
    p = vg.CalibrationProblem(0)
    cam = p.add_camera(model, d["init_intrinsics"])
    K = len(d["init_intrinsics"])
    n_glob = {"D": 0, "ID": 1, "IDDID": 4}[chain_kind]
    globs = [p.add_transform(True, np.concatenate([RNG.uniform(-0.02, 0.02, 3), RNG.uniform(-0.03, 0.03, 3)]))
             for _ in range(n_glob)]
    seq = p.add_transform(False, d["init_poses"])
    if chain_kind == "D":
        chain, status = [(seq, 0)], [0]
    elif chain_kind == "ID":
        chain, status = [(globs[0], 1), (seq, 0)], [1, 0]
    else:
        chain = [(globs[0], 1), (globs[1], 0), (seq, 0), (globs[2], 1), (globs[3], 0)]
        status = [1, 0, 0, 1, 0]
    ds = p.add_dataset(cam, chain, board, corners)
    p.finalize()
    bases = [p.transform_offset(t, 0) for t, _ in chain]
    strides = [6 if t == seq else 0 for t, _ in chain]
    return p, ds, status, board, corners, K, bases, strides


# single-member chains (W <= 17) run on the vector-pipe kernel (vg_gram_valu.hpp: one chunk of 3 corners per lane for W <= 13,
# chunks of 2 + 1 for Mei's 17), longer chains on the matrix-core kernel; test_matrix_core_kernels_for_narrow_blocks
# repeats the narrow cases with the matrix-core kernel forced
CASES = [("eucm", 40, "D", None),      # W = 13                          (config 2 / headline shape)
         ("ucm", 17, "D", None),       # W = 12
         ("mei", 23, "D", None),       # W = 17                          (config 4 shape)
         ("eucm", 19, "ID", None),     # W = 19, stereo cam-2 shape      (config 3)
         ("mei", 21, "ID", None),      # W = 23: 276 upper-triangle entries, more than a workgroup has threads
         ("mei", 9, "IDDID", None),    # W = 41, three tiles, chain of 5
         ("eucm", 11, "D", 7),         # odd N < wave: image rows end mid-MFMA group
         ("ucm", 6, "ID", 65),         # N = 65: second 64-corner tile holds one corner
         ("mei", 3, "D", 200)]


@pytest.mark.parametrize("model,n_images,chain_kind,n_points", CASES)
def test_gram_fused_two_pass_and_sum(vg, model, n_images, chain_kind, n_points):
    import torch

    p, ds, status, board, corners, K, bases, strides = build(vg, model, n_images, chain_kind, n_points)
    W = p.gram_width(ds)
    assert W == K + 6 * len(status) + 1
    pv = p.get_parameters()
    Gref = oracle_grams(model, status, board, corners, pv, 0, bases, strides, np.arange(n_images))
    gram, gsum = p.alloc_gram(ds)
    gram.fill_(float("nan"))
    p.prepare()
    p.gram_fused(ds, gram)
    p.gram_sum(ds, gram, gsum)
    p.synchronize()
    G = gram.cpu().numpy()
    assert_gram_parity(G, Gref, "fused")
    # two-pass over the materialised rows gives the same matrices (same contraction order)
    res, ji, jm = p.alloc_outputs(ds)
    gram2 = torch.full_like(gram, float("nan"))
    p.evaluate_dataset(ds, res, ji, jm)
    p.gram_from_rows(ds, res, ji, jm, gram2)
    p.synchronize()
    assert_gram_parity(gram2.cpu().numpy(), Gref, "two-pass")
    # same contraction order; the fused kernel evaluates its rows with shared reciprocals and FMA contraction
    # (a few ulp per entry), the two-pass kernel reads the reference-order rows of the emit kernel
    G2 = gram2.cpu().numpy()
    dscale = np.sqrt(np.abs(np.einsum("bii->bi", Gref)))
    assert np.max(np.abs(G - G2) / np.maximum(dscale[:, :, None] * dscale[:, None, :], 1e-300)) <= 1e-12
    # deterministic reduction over images
    ref_sum = Gref.astype(np.longdouble).sum(axis=0).astype(np.float64)
    S_ = gsum.cpu().numpy()
    assert np.linalg.norm(S_ - ref_sum) <= TOL * np.linalg.norm(ref_sum)
    gsum2 = torch.empty_like(gsum)
    p.gram_sum(ds, gram, gsum2)
    p.synchronize()
    assert torch.equal(gsum, gsum2), "reduction must be run-to-run reproducible"
    # blocks + sum in one call (vg_dataset_gram_fused_sum: per-workgroup partials of the vector-pipe kernel, one final launch)
    gram3, gsum3 = torch.full_like(gram, float("nan")), torch.full_like(gsum, float("nan"))
    p.gram_fused_sum(ds, gram3, gsum3)
    p.synchronize()
    assert torch.equal(gram3, gram)
    assert np.linalg.norm(gsum3.cpu().numpy() - ref_sum) <= TOL * np.linalg.norm(ref_sum)
    # cost = 1/2 r^T r (what Ceres reports) sits in the last entry
    r, _, _ = vgo.eval_dataset(vgo.MODELS[model], status, board, corners, pv, 0, bases, strides, np.arange(n_images),
                               want_jac=False)
    assert abs(S_[-1, -1] - np.sum(r.astype(np.longdouble) ** 2)) <= 1e-12 * S_[-1, -1]
    p.close()


def test_gram_with_failed_projections_matches_ceres_semantics(vg):
    """a failed EUCM projection contributes its in-band 1e15 residual pair and zero Jacobian rows
    (calib_cost_functions.cpp:66-70): r^T r picks up 2e30 per failed corner, J^T J and J^T r nothing."""
    from visgeom_amd import synthetic as S

    d = S.make_mono("eucm", 3, 2)
    poses = d["gt_poses"].copy()
    poses[1] = [0, 0, -1, 0, 0, 0]
    p = vg.CalibrationProblem(0)
    cam = p.add_camera("eucm", d["gt_intrinsics"])
    seq = p.add_transform(False, poses)
    ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"])
    p.finalize()
    gram, gsum = p.alloc_gram(ds)
    p.prepare()
    p.gram_fused(ds, gram)
    p.synchronize()
    Gref = oracle_grams("eucm", [0], d["board"], d["corners"], p.get_parameters(), 0, [6], [6], np.arange(3))
    G = gram.cpu().numpy()
    assert_gram_parity(G, Gref)
    assert abs(G[1][-1, -1] / (94 * 2e30) - 1) < 1e-12
    p.close()


def test_full_size_10k_gram_properties(vg):
    """config-size check: 10 k images.  Oracle comparison on a strided subset of images (the long-double
    Gram is the slow part), plus size-independent properties over all of them: exact symmetry, the sum
    kernel equals a float64 torch sum to rounding, and permutation of images permutes the blocks bit for bit."""
    import torch

    from visgeom_amd import synthetic as S

    n = 10000
    d = S.make_mono("eucm", n, 1)
    perm = RNG.permutation(n).astype(np.int32)
    p = vg.CalibrationProblem(0)
    cam = p.add_camera("eucm", d["init_intrinsics"])
    seq = p.add_transform(False, d["init_poses"])
    ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"])
    dsp = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"][perm], image_index=perm)
    p.finalize()
    gram, gsum = p.alloc_gram(ds)
    gramp, _ = p.alloc_gram(dsp)
    p.prepare()
    p.gram_fused(ds, gram)
    p.gram_fused(dsp, gramp)
    p.gram_sum(ds, gram, gsum)
    p.synchronize()
    assert torch.equal(gram, gram.transpose(1, 2))
    assert torch.equal(gramp, gram[torch.as_tensor(perm.astype(np.int64), device=gram.device)])
    tsum = gram.sum(dim=0)
    assert torch.linalg.norm(gsum - tsum) <= 1e-13 * torch.linalg.norm(tsum)
    sub = np.arange(0, n, 97)
    pv = p.get_parameters()
    Gref = oracle_grams("eucm", [0], d["board"], d["corners"][sub], pv, 0, [6], [6], sub)
    assert_gram_parity(gram.cpu().numpy()[sub], Gref, "10k subset")
    p.close()


def test_matrix_core_kernels_for_narrow_blocks(vg):
    """vg_debug_set("gram_force_mfma", 1) sends the single-member chains to vg_gram_fused_kernel as well -- one 16 x 16 tile
    for W <= 16, the RCOL variant (16 Jacobian columns in the tile, residual column on the lanes) for Mei's W = 17: the same
    cases, the same bars"""
    from visgeom_amd import capi

    capi.debug_set("gram_force_mfma", 1)
    try:
        for c in CASES:
            test_gram_fused_two_pass_and_sum(vg, *c)
        test_gram_with_failed_projections_matches_ceres_semantics(vg)
    finally:
        capi.debug_set("gram_force_mfma", 0)
    with pytest.raises(capi.VisgeomError):
        capi.debug_set("no_such_hook", 1)


# The persistent form of the direct kernel (vg_gram_valu_pers_kernel: resident workgroups, the chain walk once per workgroup
# and chunk, pairs taken from a counter, per-pair totals added in pair order) against the one-shot kernel, through the hook
# `gram_persistent` (1 = never, 2 / 3 = its four-wave / eight-wave shape whenever it applies): the per-image blocks are the SAME
# arithmetic -- bit-identical --, the sums differ by the order of additions only, and two runs of a persistent shape give the
# same bits (which wave took which pair must not show).
@pytest.mark.parametrize("model,n_images,n_points", [("eucm", 41, None),      # 21 pairs, the last one half empty
                                                     ("ucm", 64, None),
                                                     ("eucm", 9, 100),        # ragged board: one full chunk + a 4-corner remainder
                                                     ("eucm", 40001, None),   # 20 001 pairs on 512 workgroups: two chunks each
                                                     ("mei", 37, None),       # 17-wide rows: two corners of a lane, then the third
                                                     ("mei", 30001, None)])   # chunks of 24 pairs: two (four-wave shape) / five (eight-wave shape) per workgroup
def test_persistent_direct_kernel_equals_the_one_shot_kernel(vg, model, n_images, n_points):
    import torch
    from visgeom_amd import capi

    p, ds, status, board, corners, K, bases, strides = build(vg, model, n_images, "D", n_points)
    W = K + 7
    out = {}
    try:
        for name, hook in (("one-shot", 1), ("persistent", 2), ("persistent again", 2), ("eight waves", 3), ("eight waves again", 3)):
            capi.debug_set("gram_persistent", hook)
            gram, gsum = p.alloc_gram(ds)
            gram.fill_(float("nan"))
            gsum.fill_(float("nan"))
            p.prepare()
            p.gram_fused_sum(ds, gram, gsum)
            p.synchronize()
            out[name] = (gram.cpu().numpy().copy(), gsum.cpu().numpy().copy())
    finally:
        capi.debug_set("gram_persistent", 0)
    g0, s0 = out["one-shot"]
    g1, s1 = out["persistent"]
    g2, s2 = out["persistent again"]
    g3, s3 = out["eight waves"]
    g4, s4 = out["eight waves again"]
    assert g0.shape[0] == n_images and np.isfinite(g1).all() and np.isfinite(s1).all()
    assert np.array_equal(g0, g1), "per-image blocks must be bit-identical"
    assert np.array_equal(g1, g2) and np.array_equal(s1, s2), "the persistent form must not depend on which wave took which pair"
    assert np.array_equal(g0, g3) and np.array_equal(g3, g4) and np.array_equal(s3, s4)
    scale = np.sqrt(np.abs(np.outer(np.diag(s0.reshape(W, W)), np.diag(s0.reshape(W, W)))))
    assert np.max(np.abs(s1.reshape(W, W) - s0.reshape(W, W)) / np.maximum(scale, 1e-300)) <= 1e-12
    assert np.max(np.abs(s3.reshape(W, W) - s0.reshape(W, W)) / np.maximum(scale, 1e-300)) <= 1e-12
    if n_images <= 100:   # and both equal the long-double Gram of the oracle's rows
        pv = p.get_parameters()
        Gref = oracle_grams(model, status, board, corners, pv, 0, bases, strides, np.arange(n_images))
        assert_gram_parity(g1.reshape(n_images, W, W), Gref, "persistent")


def test_persistent_kernel_with_an_image_to_sequence_map(vg):
    """A dataset whose images are a shuffled subset of the pose sequence (image_index): the walker wave of the persistent kernel
    follows the map as the one-shot kernel's walkers do -- same blocks bit for bit in both shapes, and equal to the oracle."""
    from visgeom_amd import capi
    from visgeom_amd import synthetic as S

    n_seq, n_img = 90, 61
    d = S.make_mono("eucm", n_seq, 3)
    pick = RNG.permutation(n_seq)[:n_img]
    p = vg.CalibrationProblem(0)
    cam = p.add_camera("eucm", d["init_intrinsics"])
    seq = p.add_transform(False, d["init_poses"])
    ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"][pick], image_index=pick)
    p.finalize()
    W = len(d["init_intrinsics"]) + 7
    out = {}
    try:
        for name, hook in (("one-shot", 1), ("four waves", 2), ("eight waves", 3)):
            capi.debug_set("gram_persistent", hook)
            gram, gsum = p.alloc_gram(ds)
            gram.fill_(float("nan"))
            p.prepare()
            p.gram_fused_sum(ds, gram, gsum)
            p.synchronize()
            out[name] = (gram.cpu().numpy().copy(), gsum.cpu().numpy().copy())
    finally:
        capi.debug_set("gram_persistent", 0)
    assert np.array_equal(out["one-shot"][0], out["four waves"][0]) and np.array_equal(out["one-shot"][0], out["eight waves"][0])
    for name in ("four waves", "eight waves"):
        assert np.allclose(out[name][1], out["one-shot"][1], rtol=1e-12, atol=0.)
    Gref = oracle_grams("eucm", [0], d["board"], d["corners"][pick], p.get_parameters(), 0, [p.transform_offset(seq, 0)], [6], pick)
    assert_gram_parity(out["eight waves"][0].reshape(n_img, W, W), Gref, "persistent, mapped images")


def test_persistent_kernel_on_failed_projections_and_missing_corners(vg):
    """Boards behind the camera (the in-band 1e15 of calib_cost_functions.cpp:66-70) and NaN observations go through the persistent
    kernel exactly as through the one-shot kernel: every block bit for bit, NaNs in the same places."""
    from visgeom_amd import capi
    from visgeom_amd import synthetic as S

    n = 70
    d = S.make_mono("eucm", n, 5)
    poses = d["init_poses"].copy()
    poses[[3, 40], 2] = -np.abs(poses[[3, 40], 2]) - 3.      # behind the camera ...
    intr = d["init_intrinsics"].copy()
    intr[0] = 0.3                                             # ... of a model (alpha < 1/2) that cannot see there: every projection of these images fails
    corners = d["corners"].copy()
    corners[5, 10, 0] = np.nan
    corners[17, 95, :] = np.nan
    p = vg.CalibrationProblem(0)
    cam = p.add_camera("eucm", intr)
    seq = p.add_transform(False, poses)
    ds = p.add_dataset(cam, [(seq, 0)], d["board"], corners)
    p.finalize()
    out = {}
    try:
        for name, hook in (("one-shot", 1), ("four waves", 2), ("eight waves", 3)):
            capi.debug_set("gram_persistent", hook)
            gram, gsum = p.alloc_gram(ds)
            gram.fill_(0.)
            p.prepare()
            p.gram_fused_sum(ds, gram, gsum)
            p.synchronize()
            out[name] = (gram.cpu().numpy().copy(), gsum.cpu().numpy().copy())
    finally:
        capi.debug_set("gram_persistent", 0)
    g0 = out["one-shot"][0].reshape(n, -1)
    assert np.isnan(g0[5]).any() and np.isnan(g0[17]).any() and np.isfinite(g0[4]).all()
    assert np.abs(g0[3]).max() > 1e29 and np.abs(g0[40]).max() > 1e29      # 96 x 2 residuals of 1e15, squared
    for name in ("four waves", "eight waves"):
        assert np.array_equal(out[name][0], out["one-shot"][0], equal_nan=True), name
        assert np.array_equal(np.isnan(out[name][1]), np.isnan(out["one-shot"][1]))
