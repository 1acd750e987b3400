"""The library that ships (python -m visgeom_amd._build --production: no VG_DEBUG_HOOKS, vg_debug_set not exported) against the
hooks build every other test drives: the same bits from every product pass -- emit rows (single and merged launches, all three
camera models, chains of one and two members), fused Gram blocks and their sums, the second-pass Gram, and the LM solution.
(The whole GPU suite also runs against it: VISGEOM_AMD_LIBRARY=production python -m pytest tests -m gpu; tools/production_check.sh.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _dump(tmp_path, which):
    env = dict(os.environ)
    env.pop("VISGEOM_AMD_LIBRARY", None)
    if which == "production":
        env["VISGEOM_AMD_LIBRARY"] = "production"
    path = str(tmp_path / ("%s.npz" % which))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "production_dump.py"), path], env=env, cwd=ROOT)
    return np.load(path)


def test_production_library_gives_the_same_bits(tmp_path):
    from visgeom_amd import _build

    assert os.path.exists(_build.PRODUCTION_LIB), "python -m visgeom_amd._build --production (or __graft_entry__.build())"
    a, b = _dump(tmp_path, "hooks"), _dump(tmp_path, "production")
    assert int(a["has_hooks"][0]) == 1 and int(b["has_hooks"][0]) == 0
    assert str(a["library"][0]) != str(b["library"][0]) and "production" in str(b["library"][0])
    keys = sorted(k for k in a.files if k not in ("has_hooks", "library"))
    assert keys == sorted(k for k in b.files if k not in ("has_hooks", "library")) and len(keys) > 40
    for k in keys:
        x, y = a[k], b[k]
        assert x.shape == y.shape, k
        assert x.tobytes() == y.tobytes(), "%s differs between the hooks and the production library (max abs diff %g)" % (
            k, float(np.nanmax(np.abs(x - y))))
    assert np.isfinite(a["c2_solution"]).all()


def test_smoke_on_the_production_library():
    env = dict(os.environ, VISGEOM_AMD_LIBRARY="production")
    out = subprocess.check_output([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], env=env, cwd=ROOT, text=True)
    assert "smoke ok" in out
