"""BASELINE config 5 as a parity / solve case: mixed UCM / EUCM / EUCM / Mei rig, one shared board, three global
transforms used INVERSE + one per-frame pose: 45 global columns, Gram widths 12 / 19 / 19 / 23."""
import numpy as np
import pytest

from oracle import vgo
from tests.parity import assert_block_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vg():
    import torch

    assert torch.cuda.is_available()
    import visgeom_amd

    return visgeom_amd


def build_rig(vg, r, use_gt=False):
    p = vg.CalibrationProblem(0)
    key = "gt_" if use_gt else "init_"
    cams = [p.add_camera(m, r[key + "intrinsics"][k]) for k, m in enumerate(r["models"])]
    x1k = [p.add_transform(True, r[key + "xi1k"][k]) for k in range(3)]
    seq = p.add_transform(False, r[key + "poses"])
    dss = [p.add_dataset(cams[0], [(seq, 0)], r["board"], r["corners"][0])]
    for k in range(3):
        dss.append(p.add_dataset(cams[k + 1], [(x1k[k], 1), (seq, 0)], r["board"], r["corners"][k + 1]))
    p.finalize()
    return p, cams, x1k, seq, dss


def test_rig_parity_gram_and_solve(vg):
    from visgeom_amd import synthetic as S

    n = 120
    r = S.make_rig(n, sigma=0.0)
    p, cams, x1k, seq, dss = build_rig(vg, r)
    pv = p.get_parameters()
    assert p.num_parameters == 5 + 6 + 6 + 10 + 18 + 6 * n
    # residual / Jacobian parity of every dataset at the perturbed point
    p.prepare()
    for k, ds in enumerate(dss):
        model = r["models"][k]
        K = len(r["gt_intrinsics"][k])
        status = [0] if k == 0 else [1, 0]
        bases = [p.transform_offset(seq, 0)] if k == 0 else [p.transform_offset(x1k[k - 1]), p.transform_offset(seq, 0)]
        strides = [6] if k == 0 else [0, 6]
        res, ji, jm = p.alloc_outputs(ds)
        p.evaluate_dataset(ds, res, ji, jm)
        p.synchronize()
        rr, jir, jmr = vgo.eval_dataset(vgo.MODELS[model], status, r["board"], r["corners"][k], pv, p.camera_offset(cams[k]),
                                        bases, strides, np.arange(n), threads=4)
        for b in range(0, n, 7):
            assert_block_parity(res[b].cpu().numpy(), [ji[b].cpu().numpy()] + [m[b].cpu().numpy() for m in jm],
                                rr[b], [jir[b]] + [m[b] for m in jmr], r["corners"][k][b], "cam %d block %d" % (k, b))
        # normal-equation blocks of the same dataset
        gram, gsum = p.alloc_gram(ds)
        p.gram_fused(ds, gram)
        p.synchronize()
        G = gram.cpu().numpy()
        for b in range(0, n, 11):
            ref = vgo.block_gram(rr[b], jir[b], [m[b] for m in jmr])
            assert np.linalg.norm(G[b] - ref) <= 1e-10 * np.linalg.norm(ref)
    # full LM loop: GPU Gram + Schur, host Cholesky of the 45 x 45 system
    s = p.solve(max_num_iterations=100)
    x = p.get_parameters()
    print("rig", s["termination"], s["num_iterations"], "cost %.3e -> %.3e" % (s["initial_cost"], s["final_cost"]))
    assert s["num_global_columns"] == 45 and s["num_pose_blocks"] == n
    assert s["final_cost"] < 1e-12 * s["initial_cost"]
    for k in range(4):
        o = p.camera_offset(cams[k])
        gt = r["gt_intrinsics"][k]
        assert np.max(np.abs(x[o:o + len(gt)] - gt) / np.maximum(np.abs(gt), 1.0)) < 1e-6, "camera %d" % k
    for k in range(3):
        o = p.transform_offset(x1k[k])
        assert np.max(np.abs(x[o:o + 6] - r["gt_xi1k"][k])) < 1e-6
    p.close()


@pytest.mark.parametrize("n_cam,device_loop", [(8, 0), (8, 1), (4, 1)])
def test_wide_rig_more_than_63_global_columns(vg, n_cam, device_loop):
    """eight Mei cameras on one rig: G = 8 * 10 intrinsics + 7 * 6 global transforms = 122 global columns -- the reduced
    system is handled as 8 x 8 MFMA tiles (tile-pair kernel) and 8 columns per lane in the back-substitution; forced onto the
    device-resident loop it is factorised by the row-per-lane one-workgroup kernel (133 KB of LDS).  Four cameras: G = 58,
    on the device loop the entry-parallel kernel with more than 48 KB of LDS.
    Noise-free data: the solve must return the generating values."""
    from visgeom_amd import capi, synthetic as S

    n_frames = 40
    board = S.board_points()
    gts = [S.GT_MEI * (1 + 0.002 * k * np.array([1, 0, 0, 0, 0, 0, 1, 1, 0.2, 0.2])) for k in range(n_cam)]
    xi1k = [np.array([0.06 * (k % 4), 0.06 * (k // 4), 0.0, 0.004 * k, -0.003 * k, 0.002 * k]) for k in range(1, n_cam)]
    cams = [("mei", gts[0], np.eye(3), np.zeros(3))]
    for k in range(n_cam - 1):
        R = S.rodrigues(xi1k[k][3:])
        cams.append(("mei", gts[k + 1], R.T, -R.T @ xi1k[k][:3]))
    poses = S.make_poses(S.BASE_SEED + 11, n_frames, cams, board)
    X1 = np.einsum("nij,kj->nki", S.rodrigues(poses[:, 3:]), board) + poses[:, None, :3]
    p = vg.CalibrationProblem(0)
    cids = [p.add_camera("mei", S.INIT["mei"]) for _ in range(n_cam)]
    tids = [p.add_transform(True, x + 0.004) for x in xi1k]
    seq = p.add_transform(False, poses + 0.005)
    for k, (m, intr, Rc, tc) in enumerate(cams):
        uv, ok = S.project(m, intr, np.einsum("ij,nkj->nki", Rc, X1) + tc)
        assert ok.all()
        chain = [(seq, 0)] if k == 0 else [(tids[k - 1], 1), (seq, 0)]
        p.add_dataset(cids[k], chain, board, uv)
    p.finalize()
    capi.debug_set("solver_device_loop", device_loop)
    try:
        s = p.solve(max_num_iterations=300)
    finally:
        capi.debug_set("solver_device_loop", 0)
    x = p.get_parameters()
    print("wide rig", s["termination"], s["num_iterations"], "G", s["num_global_columns"], "cost %.3e -> %.3e" % (s["initial_cost"], s["final_cost"]))
    assert s["num_global_columns"] == 10 * n_cam + 6 * (n_cam - 1)
    assert s["final_cost"] < 1e-15 * s["initial_cost"]
    for k in range(n_cam):
        assert np.max(np.abs(x[10 * k:10 * k + 10] - gts[k]) / np.maximum(np.abs(gts[k]), 1.0)) < 1e-6
    for k in range(n_cam - 1):
        assert np.max(np.abs(x[10 * n_cam + 6 * k:10 * n_cam + 6 + 6 * k] - xi1k[k])) < 1e-6
    p.close()


def test_all_datasets_in_one_launch_equal_the_per_dataset_entry(vg):
    """vg_problem_evaluate merges the rig's four datasets (UCM, EUCM, EUCM, Mei; chains [D] and [I, D]) into one emit
    launch: same bits as four vg_dataset_evaluate calls, NULL blocks honoured, failed-projection counters per dataset."""
    import torch

    from visgeom_amd import synthetic as S

    n = 57
    r = S.make_rig(n, sigma=0.1)
    p, cams, x1k, seq, dss = build_rig(vg, r)
    ref = [p.alloc_outputs(ds) for ds in dss]
    got = [p.alloc_outputs(ds) for ds in dss]
    for o in got:
        o[0].fill_(float("nan"))
        o[1].fill_(float("nan"))
        for m in o[2]:
            m.fill_(float("nan"))
    p.prepare()
    for ds, o in zip(dss, ref):
        p.evaluate_dataset(ds, o[0], o[1], o[2])
    p.prepare()
    p.evaluate_all(got)
    p.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))
    assert all(p.failed_count(ds) == 0 for ds in dss)
    # constant blocks: NULL pointers for camera 1's intrinsics and the first global transform
    got2 = [p.alloc_outputs(ds) for ds in dss]
    outs = [(o[0], None if k == 1 else o[1], [None if (k == 1 and l == 0) else m for l, m in enumerate(o[2])]) for k, o in enumerate(got2)]
    got2[1][1].fill_(7.0)
    got2[1][2][0].fill_(7.0)
    p.evaluate_all(outs)
    p.synchronize()
    assert torch.all(got2[1][1] == 7.0) and torch.all(got2[1][2][0] == 7.0)
    assert torch.equal(got2[1][2][1], ref[1][2][1]) and torch.equal(got2[3][1], ref[3][1])
    # the forced chain-prep route through the same entry
    p.force_prepared_frames(True)
    p.prepare()
    p.evaluate_all(got)
    p.synchronize()
    pv = p.get_parameters()
    rr, jir, jmr = vgo.eval_dataset(vgo.MODELS[r["models"][0]], [0], r["board"], r["corners"][0], pv, p.camera_offset(cams[0]),
                                    [p.transform_offset(seq, 0)], [6], np.arange(n), threads=2)
    for b in range(0, n, 9):
        assert_block_parity(got[0][0][b].cpu().numpy(), [got[0][1][b].cpu().numpy(), got[0][2][0][b].cpu().numpy()],
                            rr[b], [jir[b], jmr[0][b]], r["corners"][0][b], "merged launch, prepared frames")
    p.close()


@pytest.mark.parametrize("n", [57, 8])
def test_all_gram_blocks_in_one_launch_equal_the_per_dataset_entry(vg, n):
    """vg_problem_gram_fused sends the rig's four datasets (W = 12, 19, 19, 23; chains [D] and [I, D]) through ONE
    vector-pipe launch: same bits as four vg_dataset_gram_fused calls (a workgroup runs the same body either way), on
    both chain routes, and ragged workgroup counts (57 images = 7 full workgroups + 1) do not leak across datasets."""
    import torch

    from visgeom_amd import synthetic as S

    r = S.make_rig(n, sigma=0.1)
    p, cams, x1k, seq, dss = build_rig(vg, r)
    for forced in (False, True):
        p.force_prepared_frames(forced)
        ref = [p.alloc_gram(ds)[0] for ds in dss]
        got = [torch.full_like(g, float("nan")) for g in ref]
        p.prepare()
        for ds, g in zip(dss, ref):
            p.gram_fused(ds, g)
        p.prepare()
        p.gram_fused_all(got)
        p.synchronize()
        for k, (a, b) in enumerate(zip(ref, got)):
            assert torch.equal(a, b), "dataset %d, forced frames %s" % (k, forced)
        # vg_problem_gram_fused_sum: the same blocks, and every dataset's fixed-order sum from ONE more launch -- bit for bit
        # the sums of the per-dataset entry
        ref_s = [p.alloc_gram(ds)[1] for ds in dss]
        got_b = [torch.full_like(g, float("nan")) for g in ref]
        got_s = [torch.full_like(t, float("nan")) for t in ref_s]
        p.prepare()
        for ds, g, t in zip(dss, ref, ref_s):
            p.gram_fused_sum(ds, g, t)
        p.prepare()
        p.gram_fused_sum_all(got_b, got_s)
        p.synchronize()
        for k in range(len(dss)):
            assert torch.equal(ref[k], got_b[k]) and torch.equal(ref_s[k], got_s[k]), "dataset %d, forced frames %s" % (k, forced)
            assert torch.allclose(got_s[k], got_b[k].sum(0), rtol=1e-12, atol=0)
    p.close()


def test_merged_gram_launch_with_datasets_it_cannot_take(vg):
    """vg_problem_gram_fused on a problem whose datasets do NOT all fit the merged launch: a 96-point EUCM set [D], a
    7-point UCM set [D] (boards of at most 32 points keep their own one-corner-per-lane launch) and a 96-point Mei set
    [I, D].  Two go together, one goes alone, the results are those of three vg_dataset_gram_fused calls; the solver
    (which then cannot use the merged partial sums) still reaches a stationary point with a lower cost."""
    import torch

    from visgeom_amd import synthetic as S

    n = 21
    r = S.make_rig(n, sigma=0.1)
    rng = np.random.default_rng(5)
    small_idx = rng.choice(96, 7, replace=False)
    p = vg.CalibrationProblem(0)
    cams = [p.add_camera(r["models"][k], r["init_intrinsics"][k]) for k in (1, 0, 3)]      # EUCM, UCM, Mei
    x13 = p.add_transform(True, r["init_xi1k"][2])
    seq = p.add_transform(False, r["init_poses"])
    # camera 0 of the rig is the UCM one with the identity mount: chains [D]; the EUCM camera is given chain [D] on its own
    # sequence so that all three datasets are valid problems of their own
    seq2 = p.add_transform(False, r["init_poses"])
    dss = [p.add_dataset(cams[0], [(seq2, 0)], r["board"], r["corners"][0]),
           p.add_dataset(cams[1], [(seq, 0)], r["board"][small_idx], r["corners"][0][:, small_idx]),
           p.add_dataset(cams[2], [(x13, 1), (seq, 0)], r["board"], r["corners"][3])]
    p.finalize()
    ref = [p.alloc_gram(ds)[0] for ds in dss]
    got = [torch.full_like(g, float("nan")) for g in ref]
    p.prepare()
    for ds, g in zip(dss, ref):
        p.gram_fused(ds, g)
    p.prepare()
    p.gram_fused_all(got)
    p.synchronize()
    for k, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), "dataset %d" % k
    # the summed entry on the same mix: two datasets from the merged launch + ONE sum launch, the third on its own route
    ref_s = [p.alloc_gram(ds)[1] for ds in dss]
    got_b = [torch.full_like(g, float("nan")) for g in ref]
    got_s = [torch.full_like(t, float("nan")) for t in ref_s]
    p.prepare()
    for ds, g, t in zip(dss, ref, ref_s):
        p.gram_fused_sum(ds, g, t)
    p.gram_fused_sum_all(got_b, got_s)
    p.synchronize()
    for k in range(3):
        assert torch.equal(ref[k], got_b[k]) and torch.equal(ref_s[k], got_s[k]), "dataset %d" % k
    s = p.solve(max_num_iterations=100)
    assert s["termination"].startswith("CONVERGENCE") and s["final_cost"] < s["initial_cost"]
    p.close()


@pytest.mark.parametrize("n_ds", [9, 13])
def test_more_datasets_than_one_merged_launch_takes(vg, n_ds):
    """The merged launches carry their datasets by value in the kernel arguments: 8 per emit launch (kEmitMultiMax), 6 per Gram
    launch (kGramMultiMax), 8 per partial-sum launch (kPartialSumMax).  A problem with 9 / 13 datasets (the rig's four cameras,
    several datasets each) is cut into 8 + 1 / 8 + 5 emit launches, 6 + 3 / 6 + 6 + a lone leftover Gram launches, 8 + 1 / 8 + 5
    sum launches -- and must give the bits of one launch per dataset (no fixed ceiling on the number of
    cameras of a rig)."""
    import torch

    from visgeom_amd import synthetic as S

    n = 19
    r = S.make_rig(n, sigma=0.1)
    rng = np.random.default_rng(3)
    p = vg.CalibrationProblem(0)
    cams = [p.add_camera(m, r["init_intrinsics"][k]) for k, m in enumerate(r["models"])]
    x1k = [p.add_transform(True, r["init_xi1k"][k]) for k in range(3)]
    seq = p.add_transform(False, r["init_poses"])
    dss, which = [], []
    for j in range(n_ds):
        k = j % 4
        chain = [(seq, 0)] if k == 0 else [(x1k[k - 1], 1), (seq, 0)]
        dss.append(p.add_dataset(cams[k], chain, r["board"], r["corners"][k] + 0.05 * rng.standard_normal(r["corners"][k].shape)))
        which.append(k)
    p.finalize()
    ref = [p.alloc_outputs(ds) for ds in dss]
    got = [p.alloc_outputs(ds) for ds in dss]
    for o in got:
        o[0].fill_(float("nan"))
        o[1].fill_(float("nan"))
        for m in o[2]:
            m.fill_(float("nan"))
    p.prepare()
    for ds, o in zip(dss, ref):
        p.evaluate_dataset(ds, o[0], o[1], o[2])
    p.prepare()
    p.evaluate_all(got)
    p.synchronize()
    for j, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), "dataset %d" % j
        assert all(torch.equal(x, y) for x, y in zip(a[2], b[2])), "dataset %d" % j
    j = n_ds - 1   # the last dataset: the one-dataset tail launch of the 9-dataset problem
    gram_ref = [p.alloc_gram(ds) for ds in dss]
    gram_got = [(torch.full_like(g, float("nan")), torch.full_like(t, float("nan"))) for g, t in gram_ref]
    p.prepare()
    for ds, (g, t) in zip(dss, gram_ref):
        p.gram_fused_sum(ds, g, t)
    p.prepare()
    p.gram_fused_sum_all([g for g, _ in gram_got], [t for _, t in gram_got])
    p.synchronize()
    for j2 in range(n_ds):
        assert torch.equal(gram_ref[j2][0], gram_got[j2][0]) and torch.equal(gram_ref[j2][1], gram_got[j2][1]), "Gram of dataset %d" % j2
    # the Gram blocks of the last dataset are those of its emitted rows (long-double Gram of the HIP rows: 1e-10)
    res, ji = got[j][0].cpu().numpy(), got[j][1].cpu().numpy()
    jm = [m.cpu().numpy() for m in got[j][2]]
    for b in (0, n - 1):
        J = np.concatenate([ji[b]] + [m[b] for m in jm] + [res[b][:, None]], axis=1).astype(np.longdouble)
        G = (J.T @ J).astype(np.float64)
        Gg = gram_got[j][0][b].cpu().numpy()
        assert np.max(np.abs(Gg - G)) <= 1e-10 * np.max(np.abs(G)), "Gram block %d of the last dataset" % b
    # the LM solve over all of them: merged launches against one launch per dataset (hook), same optimum
    from visgeom_amd import capi

    x0 = p.get_parameters()
    s1 = p.solve(max_num_iterations=60)
    x1 = p.get_parameters()
    p.set_parameters(x0)
    capi.debug_set("gram_no_merge", 1)
    try:
        s2 = p.solve(max_num_iterations=60)
    finally:
        capi.debug_set("gram_no_merge", 0)
    x2 = p.get_parameters()
    assert s1["final_cost"] < 0.05 * s1["initial_cost"]
    assert abs(s1["final_cost"] - s2["final_cost"]) <= 1e-9 * s2["final_cost"]
    assert np.max(np.abs(x1 - x2) / np.maximum(np.abs(x2), 1.0)) < 1e-6
    p.close()


@pytest.mark.parametrize("problem,loop", [("stereo", "device"), ("stereo", "host"), ("rig", "host"), ("rig", "device"), ("stereo_gaps", "device")])
def test_candidate_frames_from_the_back_substitution_equal_the_chain_prep_launch(vg, problem, loop):
    """The frames of an LM candidate are built by the back-substitution kernel that computes the candidate (no
    vg_chain_prep_multi_kernel in front of its evaluation; the reference walks the chain twice per Evaluate,
    src/calibration/calib_cost_functions.cpp:32-46,76-92).  Same walk on the same values: the WHOLE solve is bit-identical with
    the prep launch restored through the hook -- cost, iteration count and every parameter."""
    from visgeom_amd import capi, synthetic as S

    def build():
        if problem == "rig":
            return build_rig(vg, S.make_rig(70, sigma=0.1))[0]
        st = S.make_stereo(90)
        p = vg.CalibrationProblem(0)
        c1 = p.add_camera("eucm", st["init_intrinsics1"])
        c2 = p.add_camera("eucm", st["init_intrinsics2"])
        x12 = p.add_transform(True, st["init_xi12"])
        seq = p.add_transform(False, st["init_poses"])
        p.add_dataset(c1, [(seq, 0)], st["board"], st["corners1"])
        if problem == "stereo_gaps":   # the second camera saw every third pair only: blocks and sequence elements differ
            idx = np.arange(0, 90, 3, dtype=np.int32)
            p.add_dataset(c2, [(x12, 1), (seq, 0)], st["board"], st["corners2"][idx], image_index=idx)
        else:
            p.add_dataset(c2, [(x12, 1), (seq, 0)], st["board"], st["corners2"])
        p.finalize()
        return p

    out = []
    for no_fold in (0, 1):
        capi.debug_set("solver_no_fold_frames", no_fold)
        capi.debug_set("solver_device_loop" if loop == "device" else "solver_host_loop", 1)
        try:
            p = build()
            s = p.solve(max_num_iterations=60)
            out.append((s, p.get_parameters()))
            p.close()
        finally:
            capi.debug_set("solver_no_fold_frames", 0)
            capi.debug_set("solver_device_loop", 0)
            capi.debug_set("solver_host_loop", 0)
    (s_fold, x_fold), (s_prep, x_prep) = out
    assert s_fold["termination"].startswith("CONVERGENCE")
    assert s_fold["num_iterations"] == s_prep["num_iterations"] and s_fold["final_cost"] == s_prep["final_cost"]
    assert np.array_equal(x_fold, x_prep)


def test_hundreds_of_datasets_solve_to_the_optimum_of_one(vg):
    """260 datasets (one image each, one camera, one pose sequence) are one problem cut into pieces: the device-resident loop
    keeps at most 256 block widths in LDS (kLmMaxDatasets), beyond that the host-driven loop takes over -- either way the optimum
    is that of the single dataset holding the same 260 images (the reference adds one residual block per image whatever the
    grouping: src/calibration/unified_calibration.cpp:514-630)."""
    from visgeom_amd import synthetic as S

    n = 260
    d = S.make_mono("eucm", n, 2)
    one = vg.CalibrationProblem(0)
    cam = one.add_camera("eucm", d["init_intrinsics"])
    seq = one.add_transform(False, d["init_poses"])
    one.add_dataset(cam, [(seq, 0)], d["board"], d["corners"])
    one.finalize()
    s1 = one.solve(max_num_iterations=100)
    x1 = one.get_parameters()
    one.close()
    out = {}
    for n_ds in (200, 260):   # below and above the device-resident loop's limit
        many = vg.CalibrationProblem(0)
        cam = many.add_camera("eucm", d["init_intrinsics"])
        seq = many.add_transform(False, d["init_poses"])
        edges = np.linspace(0, n, n_ds + 1).astype(int)
        for j in range(n_ds):
            idx = np.arange(edges[j], edges[j + 1], dtype=np.int32)
            many.add_dataset(cam, [(seq, 0)], d["board"], d["corners"][idx], image_index=idx)
        many.finalize()
        s = many.solve(max_num_iterations=100)
        out[n_ds] = (s, many.get_parameters())
        many.close()
    for n_ds, (s, x) in out.items():
        assert s["termination"].startswith("CONVERGENCE"), (n_ds, s)
        assert abs(s["final_cost"] - s1["final_cost"]) <= 1e-9 * s1["final_cost"], (n_ds, s["final_cost"], s1["final_cost"])
        assert np.max(np.abs(x - x1) / np.maximum(np.abs(x1), 1.0)) < 1e-6, n_ds
