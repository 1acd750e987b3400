"""The parity metric of SURVEY.md section 8(c) (defined before measuring):

  residual block:  ||dr||_2 / ||proj_ref||_2 <= tol   and   |dr| <= tol * max(|r_ref|, 1 px)
  Jacobian block:  ||dJ||_F / ||J_ref||_F   <= tol   and   |dJ| <= tol * max(|J_ref|, 1e-3 * ||J_ref||_inf)
  failed-projection rows (residual pair 1e15, Jacobian rows 0) must match exactly.

tol = 1e-10 is the bar BASELINE.json's north_star states ("<= 1e-10 relative").
"""
import numpy as np

TOL = 1e-10
BIG = 1e15


def block_parity_errors(res, jacs, ref_res, ref_jacs, obs):
    """returns dict of the four normalised error figures (each must be <= 1 at tolerance TOL)."""
    res, ref_res = np.asarray(res, float).ravel(), np.asarray(ref_res, float).ravel()
    obs = np.asarray(obs, float).ravel()
    failed = ref_res == BIG
    assert np.array_equal(res == BIG, failed), "failed-projection pattern differs"
    out = {}
    ok = ~failed
    if ok.any():
        proj_ref = ref_res[ok] + obs[ok]
        dr = res[ok] - ref_res[ok]
        out["res_norm"] = np.linalg.norm(dr) / max(np.linalg.norm(proj_ref), 1e-300) / TOL
        out["res_elem"] = np.max(np.abs(dr) / np.maximum(np.abs(ref_res[ok]), 1.0)) / TOL
    if jacs is not None:
        for k, (J, Jr) in enumerate(zip(jacs, ref_jacs)):
            if Jr is None:
                assert J is None
                continue
            J, Jr = np.asarray(J, float), np.asarray(Jr, float)
            assert J.shape == Jr.shape
            if failed.any():
                assert np.all(J[failed] == 0) and np.all(Jr[failed] == 0), "failed rows must be zero"
            nf = np.linalg.norm(Jr)
            if nf == 0:
                assert np.all(J == 0)
                continue
            dJ = J - Jr
            out["jac%d_norm" % k] = np.linalg.norm(dJ) / nf / TOL
            floor = 1e-3 * np.max(np.abs(Jr))
            out["jac%d_elem" % k] = np.max(np.abs(dJ) / np.maximum(np.abs(Jr), floor)) / TOL
    return out


def assert_block_parity(res, jacs, ref_res, ref_jacs, obs, what=""):
    errs = block_parity_errors(res, jacs, ref_res, ref_jacs, obs)
    bad = {k: v for k, v in errs.items() if not (v <= 1.0)}
    assert not bad, "parity > 1e-10 %s: %s" % (what, {k: v * TOL for k, v in bad.items()})
    return errs
