"""CPU-side checks of the product boundary: the C-ABI library builds, loads, exports every symbol the
header declares, and refuses to compute without a GPU (no CPU fallback, no route through the oracle)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "visgeom_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vg_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from visgeom_amd import _build, capi

    _build.build()
    return capi.load()


def test_header_symbols_are_exported_and_bound(lib):
    from visgeom_amd import _build, capi

    syms = declared_symbols()
    assert len(syms) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", _build.LIB], text=True)
    exported = set(re.findall(r" T (vg_[a-z0-9_]+)", out))
    missing = [s for s in syms if s not in exported]
    assert not missing, "declared in the header but not exported: %s" % missing
    unbound = [s for s in syms if s not in capi.SIGNATURES]
    assert not unbound, "declared in the header but not bound in capi.py: %s" % unbound
    extra = [s for s in capi.SIGNATURES if s not in syms]
    assert not extra, "bound but not declared: %s" % extra


def test_production_library_exports_the_header_minus_the_hook_entry():
    """the library that ships (built without VG_DEBUG_HOOKS): exactly the header's symbols except vg_debug_set, from the same
    translation units, and no trace of the hook table"""
    from visgeom_amd import _build

    prod = _build.build_production()
    out = subprocess.check_output(["nm", "-D", "--defined-only", prod], text=True)
    exported = set(re.findall(r" T (vg_[a-z0-9_]+)", out))
    assert exported == set(declared_symbols()) - {"vg_debug_set"}, sorted(exported ^ (set(declared_symbols()) - {"vg_debug_set"}))
    blob = open(prod, "rb").read()
    assert b"gfx950" in blob and b"vg_emit_kernel" in blob
    assert b"inline_chain_max_bytes" not in blob and b"emit_map_window" not in blob   # the hook names are compiled out
    L = ctypes.CDLL(prod)
    assert L.vg_abi_version() == 1 and not hasattr(L, "vg_debug_set")


def test_library_is_hip_code_for_gfx950():
    from visgeom_amd import _build

    _build.build()
    blob = open(_build.LIB, "rb").read()
    assert b"gfx950" in blob, "no gfx950 code object inside the shared library"
    assert b"vg_emit_kernel" in blob and b"vg_chain_prep_multi_kernel" in blob


def test_static_facts(lib):
    assert lib.vg_abi_version() == 1
    assert [lib.vg_num_intrinsics(m) for m in (0, 1, 2, 3)] == [6, 5, 10, -1]
    lo, hi = ctypes.c_double(), ctypes.c_double()
    # eucm.h:228-246, ucm.h:199-215, mei.h:287-313
    expect = {(0, 0): (0, 1), (0, 1): (0.1, 10), (0, 2): (1, 1e5), (1, 0): (0, 3), (1, 4): (1, 1e5),
              (2, 0): (0, 3), (2, 3): (-10, 10), (2, 5): (-10, 10), (2, 6): (1, 1e5)}
    for (m, i), (l, h) in expect.items():
        assert lib.vg_intrinsic_bounds(m, i, ctypes.byref(lo), ctypes.byref(hi)) == 0
        assert (lo.value, hi.value) == (l, h)
    assert lib.vg_intrinsic_bounds(1, 5, ctypes.byref(lo), ctypes.byref(hi)) != 0


def test_no_cpu_fallback(lib):
    """Without a GPU every compute entry must fail loudly (VG_ERR_NO_DEVICE), never compute on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path cannot be exercised")
    from visgeom_amd import capi

    assert lib.vg_device_count() == 0
    h = ctypes.c_void_p()
    rc = lib.vg_problem_create(ctypes.byref(h), 0, None)
    assert rc == capi.ERR_NO_DEVICE and not h.value
    assert b"no CPU fallback" in lib.vg_last_error()
    import numpy as np

    import visgeom_amd

    with pytest.raises(capi.VisgeomError) as ei:
        visgeom_amd.GenericProjectionJac(np.zeros((4, 2)), np.ones((4, 3)), "eucm", [0])
    assert ei.value.code == capi.ERR_NO_DEVICE


def test_product_path_never_touches_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pat = re.compile(r"\boracle\b|vg_oracle|vgo\b")
    for base in ("visgeom_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    src = open(os.path.join(dirpath, f)).read()
                    hits = [ln for ln in src.splitlines() if pat.search(ln)
                            and "not the oracle" not in ln and "and not the oracle" not in ln
                            and "GPU and oracle differ" not in ln]
                    assert not hits, "%s mentions the oracle: %s" % (os.path.join(dirpath, f), hits[:3])
    out = subprocess.run(["ldd", os.path.join(ROOT, "visgeom_amd", "lib", "libvisgeom_amd.so")],
                         capture_output=True, text=True).stdout
    assert "vg_oracle" not in out


def test_oracle_header_declares_itself_test_infrastructure():
    for f in ("vg_oracle.h", "vg_oracle.c", "vgo.py"):
        head = open(os.path.join(ROOT, "oracle", f)).read()[:1500].upper()
        assert "TEST INFRASTRUCTURE" in head


def test_host_cholesky_solve(lib):
    """host-side linear algebra of the LM driver (the G x G reduced system), no GPU needed."""
    import numpy as np

    from visgeom_amd import capi

    rng = np.random.default_rng(3)
    dp = ctypes.POINTER(ctypes.c_double)
    for n in (1, 6, 18, 45):
        M = rng.normal(size=(n, n))
        A = np.ascontiguousarray(M @ M.T + n * np.eye(n))
        b = rng.normal(size=n)
        x = np.empty(n)
        assert lib.vg_host_cholesky_solve(n, A.ctypes.data_as(dp), b.ctypes.data_as(dp), x.ctypes.data_as(dp)) == 0
        assert np.max(np.abs(x - np.linalg.solve(A, b))) < 1e-10
    A = np.ascontiguousarray(np.array([[1.0, 2.0], [2.0, 1.0]]))  # indefinite
    b, x = np.ones(2), np.empty(2)
    assert lib.vg_host_cholesky_solve(2, A.ctypes.data_as(dp), b.ctypes.data_as(dp), x.ctypes.data_as(dp)) == capi.ERR_NUMERIC


def test_solve_option_defaults_mirror_the_reference(lib):
    from visgeom_amd import capi

    o = capi.SolveOptions()
    lib.vg_solve_options_init(ctypes.byref(o))
    # Solver::Options of GenericCameraCalibration::compute, unified_calibration.cpp:42-52
    assert (o.max_num_iterations, o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (1000, 1e-15, 1e-15, 1e-15)
    assert o.initial_trust_region_radius == 1e4 and o.use_bounds == 1 and not o.allreduce


def test_header_is_plain_c99(tmp_path):
    """the boundary is a C ABI: the header must compile as C99 (no C++-isms), and a C program must link against
    the shared library"""
    from visgeom_amd import _build

    src = tmp_path / "use.c"
    src.write_text('#include "visgeom_amd.h"\n#include <stdio.h>\n'
                   "int main(void) { vg_solve_options o; vg_solve_options_init(&o);\n"
                   '  printf("%d %d %d\\n", vg_abi_version(), vg_num_intrinsics(VG_MODEL_MEI), o.max_num_iterations);\n'
                   "  return 0; }\n")
    exe = tmp_path / "use"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", _build.LIB_DIR, "-lvisgeom_amd",
                           "-Wl,-rpath," + _build.LIB_DIR, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)], text=True).split()
    assert out == ["1", "10", "1000"]


def test_odometry_prior_host_block_matches_oracle(lib):
    """vg_odometry_prior_evaluate is host arithmetic (no GPU needed): same residual and Jacobians as the C
    restatement of OdometryPrior (calib_cost_functions.cpp:119-212), incl. the floors of the constructor."""
    import numpy as np

    from oracle import vgo

    rng = np.random.default_rng(11)
    dp = ctypes.POINTER(ctypes.c_double)
    P = lambda a: a.ctypes.data_as(dp)
    cases = [(0.05, 0.02, 0.3, 0.3), (0.0, 0.0, 1.0, 0.0), (0.5, 0.5, 0.01, 1.5)]   # (err_v, err_w, lambda, motion scale)
    for errV, errW, lam, scale in cases:
        o1 = rng.standard_normal(6) * 0.5
        o2 = vgo.compose(o1, rng.standard_normal(6) * scale)
        x1 = o1 + 0.05 * rng.standard_normal(6)
        x2 = o2 + 0.05 * rng.standard_normal(6)
        r, J1, J2 = np.empty(6), np.empty((6, 6)), np.empty((6, 6))
        assert lib.vg_odometry_prior_evaluate(errV, errW, lam, P(o1), P(o2), P(x1), P(x2), P(r), P(J1), P(J2)) == 0
        ro, J1o, J2o = vgo.OdometryPrior(errV, errW, lam, o1, o2).evaluate(x1, x2)
        sc = max(1.0, np.max(np.abs(J1o)))
        assert np.max(np.abs(r - ro)) <= 1e-12 * max(1.0, np.max(np.abs(ro)))
        assert np.max(np.abs(J1 - J1o)) <= 1e-12 * sc and np.max(np.abs(J2 - J2o)) <= 1e-12 * sc
        # Jacobians optional
        r2 = np.empty(6)
        assert lib.vg_odometry_prior_evaluate(errV, errW, lam, P(o1), P(o2), P(x1), P(x2), P(r2), None, None) == 0
        assert np.array_equal(r, r2)
    assert lib.vg_odometry_prior_evaluate(0.1, 0.1, 0.0, P(o1), P(o2), P(x1), P(x2), P(r), None, None) != 0


def test_comm_entries_without_a_gpu():
    """the RCCL binding is resolved at run time: the id can be drawn on a CPU-only box when librccl is installed (or
    fails cleanly when it is not); argument errors never reach RCCL"""
    import ctypes

    from visgeom_amd import capi

    L = capi.load()
    buf = ctypes.create_string_buffer(128)
    rc = L.vg_comm_unique_id(buf)
    assert rc in (capi.OK, capi.ERR_STATE, capi.ERR_HIP), L.vg_last_error()
    assert L.vg_comm_unique_id(None) == capi.ERR_INVALID_ARGUMENT
    h = ctypes.c_void_p()
    assert L.vg_comm_create(ctypes.byref(h), buf.raw, 2, 2, 0) == capi.ERR_INVALID_ARGUMENT
    assert L.vg_comm_create(ctypes.byref(h), buf.raw, 0, 0, 0) == capi.ERR_INVALID_ARGUMENT
    assert L.vg_comm_size(None) == -1 and L.vg_comm_rank(None) == -1
    assert L.vg_comm_allreduce_sum(None, None, 4, None) == capi.ERR_INVALID_ARGUMENT
    L.vg_comm_destroy(None)


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` must not need a launcher: it re-executes itself under torch.distributed.run with one
    rank per GPU on 127.0.0.1"""
    import importlib.util
    import subprocess

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_odometry_cost_host_block_matches_oracle(lib):
    """vg_odometry_cost_evaluate (host arithmetic, no GPU): OdometryCost with parameter blocks (xi1[6], xi2[6],
    [radius_left, radius_right, track_gauge]) (odometry_cost_function.cpp:147-266) against the C restatement, which
    tests/test_oracle_mpmath.py holds to 1e-12 of a 50-digit evaluation of the same formulas."""
    import numpy as np

    from oracle import vgo

    rng = np.random.default_rng(23)
    dp = ctypes.POINTER(ctypes.c_double)
    P = lambda a: a.ctypes.data_as(dp)
    for errV, errW, lam, n_steps, turn in [(0.05, 0.02, 0.3, 1, 0.2), (0.0, 0.0, 1.0, 7, 0.0), (0.3, 0.3, 0.02, 25, 1.0)]:
        prior = np.array([0.05, 0.052, 0.31])
        dq = np.ascontiguousarray(0.4 + 0.1 * rng.standard_normal((n_steps, 2)) + turn * np.array([-0.1, 0.1]))
        blk = vgo.OdometryCost(errV, errW, lam, dq, prior)
        x1 = rng.standard_normal(6) * 0.5
        x2 = vgo.compose(x1, blk.zeta) + 0.03 * rng.standard_normal(6)
        intr = prior * (1 + 0.05 * rng.standard_normal(3))
        z, r, J1, J2, J3 = np.empty(6), np.empty(6), np.empty((6, 6)), np.empty((6, 6)), np.empty((6, 3))
        assert lib.vg_odometry_cost_evaluate(errV, errW, lam, n_steps, P(dq), P(prior), P(x1), P(x2), P(intr), P(z), P(r), P(J1), P(J2), P(J3)) == 0
        ro, J1o, J2o, J3o = blk.evaluate(x1, x2, intr)
        assert np.max(np.abs(z - blk.zeta)) <= 1e-14 * max(1.0, np.max(np.abs(blk.zeta)))
        assert np.max(np.abs(r - ro)) <= 1e-12 * max(1.0, np.max(np.abs(ro)))
        for J, Jo in ((J1, J1o), (J2, J2o), (J3, J3o)):
            assert np.max(np.abs(J - Jo)) <= 1e-12 * max(1.0, np.max(np.abs(Jo)))
        r2 = np.empty(6)
        assert lib.vg_odometry_cost_evaluate(errV, errW, lam, n_steps, P(dq), P(prior), P(x1), P(x2), P(intr), None, P(r2), None, None, None) == 0
        assert np.array_equal(r, r2)
    assert lib.vg_odometry_cost_evaluate(0.1, 0.1, 0.0, 1, P(dq), P(prior), P(x1), P(x2), P(intr), None, P(r), None, None, None) != 0
    assert lib.vg_odometry_cost_evaluate(0.1, 0.1, 1.0, 0, P(dq), P(prior), P(x1), P(x2), P(intr), None, P(r), None, None, None) != 0
