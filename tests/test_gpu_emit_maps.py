"""The emit kernels' tile maps (vg_kernels.hpp: xcd_contiguous_block, xcd_window_block).  Which workgroup writes which 256-observation
tile is a launch-time choice of the host by output size (contiguous eighths per XCD up to 1.2 GB, windows of 8 x 16 tiles beyond):
any map must be a bijection of the tiles, i.e. give the same rows bit for bit.  Here: (1) every window width through the hook on
ragged tile counts -- whole windows, a partial last window, fewer tiles than one window, a tile count that is not a multiple of 8 --
for the single-dataset and the merged launch; (2) a launch that is large enough (100 k images, 2.15 GB) to take the windowed map and
non-temporal stores BY ITSELF -- the route of the 100 k ... 1 M image rows of bench.py's emit_sweep -- held to the oracle on a strided
subset of its blocks and, on all of them, to the rows the same images give in small launches (this one also runs on the production
library, which has no hook)."""
import numpy as np
import pytest

from oracle import vgo
from tests.parity import assert_block_parity

pytestmark = pytest.mark.gpu


def _rows(p, ds):
    res, ji, jm = p.alloc_outputs(ds)
    p.prepare()
    p.evaluate_dataset(ds, res, ji, jm)
    p.synchronize()
    return [res.cpu().numpy(), ji.cpu().numpy()] + [m.cpu().numpy() for m in jm]


@pytest.mark.parametrize("n_images", [3, 43, 683, 1366])   # 2, 17, 257, 513 tiles of 256 observations
def test_every_window_width_gives_the_same_rows(n_images):
    from visgeom_amd import CalibrationProblem, capi, synthetic as S

    d = S.make_mono("mei", n_images, 7)
    p = CalibrationProblem(0)
    cam = p.add_camera("mei", d["init_intrinsics"])
    seq = p.add_transform(False, d["init_poses"])
    ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"])
    p.finalize()
    try:
        capi.debug_set("emit_map_window", -1)
        ref = _rows(p, ds)
        for W in (1, 2, 3, 5, 16, 31, 64, 4096):
            capi.debug_set("emit_map_window", W)
            for forced in (False, True):   # the kernel walks the chain itself / reads prepared frames
                p.force_prepared_frames(forced)
                got = _rows(p, ds)
                if forced:   # other frames route: rounding-level differences in the frames, not in the map
                    for a, b in zip(got, ref):
                        assert np.allclose(a, b, rtol=1e-11, atol=1e-9), W
                else:
                    for a, b in zip(got, ref):
                        assert a.tobytes() == b.tobytes(), "W = %d changes the rows" % W
            p.force_prepared_frames(False)
    finally:
        capi.debug_set("emit_map_window", 0)
        p.close()


def test_merged_launch_under_every_window_width():
    from visgeom_amd import benchlib, capi

    p, dss, gt, name = benchlib.build(5, 0, 301)   # rig: four datasets, three camera models, 113 tiles each (not a multiple of 8)
    f = benchlib.passes(p, dss)
    outs, _ = f["keep"]

    def snap():
        f["emit"]()
        p.synchronize()
        return [t.cpu().numpy().copy() for res, ji, jm in outs for t in [res, ji] + list(jm)]

    try:
        capi.debug_set("emit_map_window", -1)
        ref = snap()
        for W in (1, 3, 16, 200):
            capi.debug_set("emit_map_window", W)
            for a, b in zip(snap(), ref):
                assert a.tobytes() == b.tobytes(), "W = %d changes the merged launch's rows" % W
    finally:
        capi.debug_set("emit_map_window", 0)
        p.close()


def test_a_launch_beyond_1p2_GB_takes_the_windowed_map_and_still_equals_the_oracle():
    import torch

    from visgeom_amd import CalibrationProblem, synthetic as S

    n = 100000   # 100 000 x 96 x 224 B = 2.15 GB: windows (>= 1.2 GB) + non-temporal stores + chain prep (> 2.0 GB), chosen by the library
    d = S.make_mono("eucm", n, 9)
    p = CalibrationProblem(0)
    cam = p.add_camera("eucm", d["init_intrinsics"])
    seq = p.add_transform(False, d["init_poses"])
    ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"])
    p.finalize()
    assert p._lib.vg_dataset_single_launch(p._h, ds) == 0
    res, ji, jm = p.alloc_outputs(ds)
    res.fill_(float("nan")); ji.fill_(float("nan")); jm[0].fill_(float("nan"))
    p.prepare()
    p.evaluate_dataset(ds, res, ji, jm)
    p.synchronize()
    assert p.failed_count(ds) == 0
    # every row was written exactly where it belongs: no NaN left, and block b's rows equal those of a small problem of block b
    assert not torch.isnan(res).any() and not torch.isnan(ji).any() and not torch.isnan(jm[0]).any()
    pv = p.get_parameters()
    pick = np.arange(0, n, 997)
    r_ref, ji_ref, jm_ref = vgo.eval_dataset(vgo.MODEL_EUCM, [0], d["board"], d["corners"][pick], np.concatenate([pv[:6], pv[6:].reshape(n, 6)[pick].ravel()]),
                                             0, [6], [6], np.arange(pick.size), threads=4)
    R, JI, JM = res[pick].cpu().numpy(), ji[pick].cpu().numpy(), jm[0][pick].cpu().numpy()
    for k, b in enumerate(pick):
        assert_block_parity(R[k], [JI[k], JM[k]], r_ref[k], [ji_ref[k], jm_ref[0][k]], d["corners"][b], "block %d of the 2.15 GB launch" % b)
    # the same images in a small launch (contiguous eighths, plain stores, prepared frames forced): same bits
    lo, hi = 41000, 41900
    q = CalibrationProblem(0)
    qc = q.add_camera("eucm", d["init_intrinsics"])
    qs = q.add_transform(False, d["init_poses"][lo:hi])
    qd = q.add_dataset(qc, [(qs, 0)], d["board"], d["corners"][lo:hi])
    q.finalize()
    q.force_prepared_frames(True)
    small = _rows(q, qd)
    for a, b in zip(small, [res[lo:hi].cpu().numpy(), ji[lo:hi].cpu().numpy(), jm[0][lo:hi].cpu().numpy()]):
        assert a.tobytes() == b.tobytes()
    q.close()
    p.close()
