"""BASELINE config 1 (plumbing): a calibration JSON in the reference's schema with a 12 x 8 board and 20 synthetic
images supplied as pre-extracted corners, through the front-end mirror of GenericCameraCalibration: pose
initialisation (estimateInitialGrid + refinement), solve, report, image_error_<i>.txt."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch

    assert torch.cuda.is_available()


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - b) / np.maximum(np.abs(b), 1.0))


@pytest.mark.parametrize("model", ["eucm", "ucm", "mei"])
def test_config1_noise_free_from_scratch_poses(gpu, tmp_path, model):
    """no pose priors: every pose is initialised from the four outer corners (unified_calibration.cpp:1066-1135),
    refined with the intrinsics fixed, then everything is solved -- must land on the generating values"""
    from visgeom_amd import synthetic as S
    from visgeom_amd.calibration import GenericCameraCalibration

    d = S.make_mono(model, 20, 0, sigma=0.0)
    path = S.write_calibration_json(str(tmp_path), d, model, prior=False, init=True, skip=(7,))
    c = GenericCameraCalibration()
    c.addResiduals(path)
    init = c.transform("xiCamBoard")
    assert init.shape == (20, 6)
    assert np.array_equal(init[7], [0, 0, 1, 0, 0, 0])          # skipped image keeps the placeholder pose (:478)
    report = c.compute(max_num_iterations=200)
    assert "Intrinsic parameters :" in report and "Sequence : xiCamBoard" in report and "Global extrinsic parameters :" in report
    print(model, c.summary["termination"], c.summary["num_iterations"], "%.3e -> %.3e" % (c.summary["initial_cost"], c.summary["final_cost"]))
    assert rel(c.intrinsics("cam"), d["gt_intrinsics"]) < 1e-6
    poses = c.transform("xiCamBoard")
    keep = [i for i in range(20) if i != 7]
    assert np.max(np.abs(poses[keep] - d["gt_poses"][keep])) < 1e-6
    c.close()


def test_config1_noisy_report_and_residual_file(gpu, tmp_path):
    from visgeom_amd import synthetic as S
    from visgeom_amd.calibration import GenericCameraCalibration

    d = S.make_mono("eucm", 20, 0, sigma=0.1)
    path = S.write_calibration_json(str(tmp_path), d, "eucm", prior=False, init=True, skip=(3,))
    c = GenericCameraCalibration()
    c.addResiduals(path)
    c.compute(max_num_iterations=200)
    assert rel(c.intrinsics("cam"), d["gt_intrinsics"]) < 5e-3   # 0.1 px noise, 19 images
    out = tmp_path / "image_error_0.txt"
    sigma, outliers = c.writeImageResidual(0, out, n_images=20)
    rows = np.loadtxt(out)
    assert rows.shape == (19 * 96, 10)                            # err(2) proj(2) t(3) r(3), skipped image absent
    assert sigma[3] == 0 and np.all((sigma[np.arange(20) != 3] > 0.07) & (sigma[np.arange(20) != 3] < 0.2))
    assert outliers == 0
    # err = detected - projected, printed with 6 significant digits
    det = np.delete(d["corners"], 3, axis=0).reshape(-1, 2)
    assert np.max(np.abs(rows[:, 0:2] - (det - rows[:, 2:4]))) < 6e-3  # proj has 6 significant digits: +-0.005 px at 1000 px
    poses = np.delete(c.transform("xiCamBoard"), 3, axis=0)
    assert np.max(np.abs(rows[::96, 4:10] - poses)) < 1e-5 * np.max(np.abs(poses))
    c.close()


def test_calib_cli_end_to_end(gpu, tmp_path):
    from visgeom_amd import _build, synthetic as S

    d = S.make_mono("eucm", 12, 0, sigma=0.0)
    path = S.write_calibration_json(str(tmp_path), d, "eucm", prior=False, init=True)
    r = subprocess.run([_build.CLI, path], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Model : EUCM" in r.stdout and "Intrinsic parameters :" in r.stdout and "Solver Summary" in r.stdout
    line = [l for l in r.stdout.splitlines() if l.startswith("cam : ")][0]
    vals = np.array([float(v) for v in line.split(":")[1].split()])
    assert rel(vals, d["gt_intrinsics"]) < 1e-5                    # printed with 6 significant digits
    assert os.path.exists(tmp_path / "image_error_0.txt")
