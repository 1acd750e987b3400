import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- the production library (VISGEOM_AMD_LIBRARY=production python -m pytest tests -m gpu) -----------------------------------
# The library that ships is built without the debug hooks (python -m visgeom_amd._build --production).  The whole suite runs
# against it unchanged; a test that needs a hook (forces an alternative route for an A/B or a bit-equality check) skips.
if os.environ.get("VISGEOM_AMD_LIBRARY"):
    from visgeom_amd import capi as _capi

    _dir = os.path.dirname(os.path.abspath(_capi.lib_path()))
    # the C++ hosts and the `calib` program the tests start find libvisgeom_amd.so through DT_RUNPATH, which LD_LIBRARY_PATH precedes
    os.environ["LD_LIBRARY_PATH"] = _dir + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")


def _skip_without_hooks(outcome):
    from visgeom_amd import capi

    exc = outcome.excinfo
    if exc is not None and isinstance(exc[1], capi.NoDebugHooks):
        outcome.force_exception(pytest.skip.Exception("needs a debug hook; the production library has none"))


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_setup(item):
    _skip_without_hooks((yield))


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    _skip_without_hooks((yield))


def pytest_report_header(config):
    from visgeom_amd import capi

    return "visgeom_amd library: %s" % capi.lib_path()
