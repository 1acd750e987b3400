"""Pins oracle/vg_oracle.c to outputs of the REFERENCE's own GenericProjectionJac::Evaluate
(src/calibration/calib_cost_functions.cpp:28-117) -- tests/golden/ref_eval_block.json, produced by tools/gen_ref_fixtures.py
from oracle/_ref/libvg_ref.so (oracle/build_ref.sh: the reference's sources compiled where they lie against REAL Eigen3 and
Ceres).  The image this project is built in has neither library, so the fixture file does not exist
there and the comparison SKIPS -- parity stays "unpinned" until somebody runs the two commands in a container that has them.
What runs everywhere: the recipe exits 77 instead of building against stand-ins, and the case matrix is well formed."""
import importlib.util
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import vgo
from tests.parity import assert_block_parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "ref_eval_block.json")


def _generator():
    spec = importlib.util.spec_from_file_location("gen_ref_fixtures", os.path.join(ROOT, "tools", "gen_ref_fixtures.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_the_recipe_builds_or_reports_absent_headers_never_stand_ins():
    r = subprocess.run(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode in (0, 77), r.stdout + r.stderr
    built = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libvg_ref.so"))
    assert built == (r.returncode == 0)
    # no Eigen / ceres look-alike headers anywhere in the repository
    for d, _, files in os.walk(ROOT):
        if "/.git" in d or "gpurun_out" in d:
            continue
        assert not (os.path.basename(d) in ("Eigen", "ceres") and files), "stand-in headers at %s" % d


def test_the_case_matrix_covers_models_chains_and_branch_edges():
    cs = _generator().cases()
    names = [c["name"] for c in cs]
    assert len(set(names)) == len(names)
    for model in (0, 1, 2):
        for chain in ("D", "ID", "DID", "IDDID"):
            for edge in ("generic", "rot0", "rot1e-6", "rot1e-5-ulp", "rot>pi", "behind"):
                assert any(n.startswith("m%d_%s_%s_" % (model, chain, edge)) for n in names), (model, chain, edge)
    # the oracle accepts every case (no reference needed for that)
    for c in cs[:40]:
        res, J = vgo.eval_block(c["model"], c["status"], np.array(c["board"]), np.array(c["obs"]),
                                [np.array(c["intrinsics"])] + [np.array(m) for m in c["members"]], jac_mask=c["jac_mask"])
        assert res.shape == (2 * len(c["board"]),)


@pytest.mark.skipif(not os.path.exists(FIX), reason="tests/golden/ref_eval_block.json absent: no real Eigen3 / Ceres in this image "
                    "(bash oracle/build_ref.sh && python tools/gen_ref_fixtures.py produce it)")
def test_oracle_equals_the_reference_outputs():
    data = json.load(open(FIX))
    for c in data["cases"]:
        assert c["ref_return"] == 1     # Evaluate always returns true (calib_cost_functions.cpp:116)
        ref_res = np.array([float.fromhex(v) for v in c["ref_residual"]])
        ref_J = [None if j is None else np.array([float.fromhex(v) for v in j]).reshape(2 * len(c["board"]), -1) for j in c["ref_jacobians"]]
        res, J = vgo.eval_block(c["model"], c["status"], np.array(c["board"]), np.array(c["obs"]),
                                [np.array(c["intrinsics"])] + [np.array(m) for m in c["members"]], jac_mask=c["jac_mask"])
        assert_block_parity(res, J, ref_res, ref_J, np.array(c["obs"]), what=c["name"])
