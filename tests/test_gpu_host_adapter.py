"""A COMPILED C++ host on the drop-in boundary: tests/host/ceres_like_host.cpp includes
integration/generic_projection_jac.h -- the adapter that replaces the reference's `struct GenericProjectionJac`
(include/calibration/calib_cost_functions.h:27-62, created at src/calibration/unified_calibration.cpp:532) -- verbatim,
derives it from a ceres::CostFunction stand-in and calls Evaluate in Ceres' pattern (state arrays, cost-only calls,
Jacobian calls, vg_block_group_invalidate after the "solve", a last Evaluate on user memory with constant intrinsics).
Everything the virtual call returned is compared with the oracle at the SURVEY 8(c) bar (1e-10).

The CPU half (`-m "not gpu"`): the host compiles against include/visgeom_amd.h as C++11, and the snippet printed in
INTEGRATION.md section 1 is the adapter file itself, not a paraphrase."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import vgo
from tests.parity import assert_block_parity
from visgeom_amd import _build, synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_SRC = os.path.join(ROOT, "tests", "host", "ceres_like_host.cpp")
ADAPTER = os.path.join(ROOT, "integration", "generic_projection_jac.h")
MODELS = {"eucm": 0, "ucm": 1, "mei": 2}


def build_host(tmp_path):
    _build.build()
    exe = os.path.join(str(tmp_path), "ceres_like_host")
    cmd = ["g++", "-O2", "-std=c++11", "-Wall", "-Werror", HOST_SRC, "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "integration"), "-L" + _build.LIB_DIR, "-lvisgeom_amd", "-Wl,-rpath," + _build.LIB_DIR,
           "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_the_adapter_compiles_as_cxx11_against_the_c_abi(tmp_path):
    assert os.path.exists(build_host(tmp_path))


def test_integration_md_prints_the_adapter_file_itself():
    """the code block of INTEGRATION.md section 1 that defines GenericProjectionJac is the compiled file from its first
    `#include <visgeom_amd.h>` to the end, character for character"""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = open(ADAPTER).read()
    body = src[src.index("#include <visgeom_amd.h>"):].rstrip()
    blocks = re.findall(r"```cpp\n(.*?)```", md, flags=re.S)
    mine = [b for b in blocks if "struct GenericProjectionJac : ceres::CostFunction" in b]
    assert len(mine) == 1, "INTEGRATION.md must print the adapter exactly once"
    assert body in mine[0], "INTEGRATION.md section 1 differs from integration/generic_projection_jac.h"


def write_case(path, model, status, is_global, board, corners, states, grouped):
    L, n, N = len(status), corners.shape[0], board.shape[0]
    h = np.zeros(17, dtype=np.int64)
    h[:7] = [MODELS[model], states[0]["intr"].size, L, N, n, len(states), int(grouped)]
    h[7:7 + L] = status
    h[12:12 + L] = [int(g) for g in is_global]
    with open(path, "wb") as f:
        f.write(h.tobytes())
        f.write(np.ascontiguousarray(board, float).tobytes())
        f.write(np.ascontiguousarray(corners, float).tobytes())
        for st in states:
            f.write(np.ascontiguousarray(st["intr"], float).tobytes())
            for l in range(L):
                f.write(np.ascontiguousarray(st["members"][l], float).tobytes())


def read_and_check(path, model, status, is_global, board, corners, states):
    """walk the output in the order the host wrote it and hold every block of every pass to the oracle"""
    L, n, N = len(status), corners.shape[0], board.shape[0]
    K = states[0]["intr"].size
    raw = np.fromfile(path, dtype=np.float64)
    pos = [0]

    def take(count, shape):
        a = raw[pos[0]:pos[0] + count].reshape(shape)
        pos[0] += count
        return a

    def params(st, i):
        return [st["intr"]] + [st["members"][l] if is_global[l] else st["members"][l][i] for l in range(L)]

    def check_pass(st, jacobians, intr_constant, what):
        for i in range(n):
            res = take(2 * N, (2 * N,))
            J = None
            if jacobians:
                J = [None if intr_constant else take(2 * N * K, (2 * N, K))] + [take(2 * N * 6, (2 * N, 6)) for _ in range(L)]
            mask = None if not intr_constant else [False] + [True] * L
            ref_res, ref_J = vgo.eval_block(vgo.MODELS[model], status, board, corners[i], params(st, i), want_jac=jacobians, jac_mask=mask)
            assert_block_parity(res, J, ref_res, ref_J if jacobians else None, corners[i], what="%s, block %d" % (what, i))

    check_pass(states[0], True, False, "initial Jacobian pass")
    for it in range(1, len(states)):
        check_pass(states[it], False, False, "candidate %d (cost only)" % it)
        check_pass(states[it], True, False, "accepted point %d" % it)
    check_pass(states[-1], True, True, "report pass on user memory, constant intrinsics")
    stats = raw[pos[0]:pos[0] + 4].view(np.int64)
    assert pos[0] + 4 == raw.size
    return {"blocks": int(stats[0]), "batched": int(stats[1]), "served": int(stats[2]), "alone": int(stats[3])}


def make_states(rng, intr0, members0, n_iter):
    out = []
    for it in range(n_iter):
        s = 0.0 if it == 0 else 1e-3
        out.append({"intr": intr0 * (1 + s * rng.standard_normal(intr0.size)),
                    "members": [m + s * rng.standard_normal(m.shape) for m in members0]})
    return out


CASES = [("eucm", "mono"), ("mei", "mono"), ("ucm", "stereo_second_camera")]


@pytest.mark.gpu
@pytest.mark.parametrize("grouped", [False, True])
@pytest.mark.parametrize("model,kind", CASES)
def test_compiled_host_through_the_adapter_equals_the_oracle(tmp_path, model, kind, grouped):
    exe = build_host(tmp_path)
    rng = np.random.default_rng(17)
    n = 23 if grouped else 5
    if kind == "mono":
        d = S.make_mono(model, n, 2)
        status, is_global = [0], [False]
        board, corners = d["board"], d["corners"]
        intr0, members0 = d["init_intrinsics"], [d["init_poses"]]
    else:   # the second camera of a stereo pair: chain [xiCam12 INVERSE (global), xiCam1Board DIRECT (sequence)]
        s = S.make_stereo(n)
        status, is_global = [1, 0], [True, False]
        board, corners = s["board"], s["corners2"]
        # the generator's second camera is EUCM; evaluate its chain with the model under test at plausible intrinsics
        intr0 = {"ucm": np.array([1.2, 307.318, 289.542, 642.617, 398.42])}.get(model, s["init_intrinsics2"])
        members0 = [s["init_xi12"], s["init_poses"]]
    states = make_states(rng, intr0, members0, 4)
    case, out = os.path.join(str(tmp_path), "case.bin"), os.path.join(str(tmp_path), "out.bin")
    write_case(case, model, status, is_global, board, corners, states, grouped)
    r = subprocess.run([exe, case, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    stats = read_and_check(out, model, status, is_global, board, corners, states)
    if grouped:
        # the candidate / accepted passes of the "solve" were answered from batched passes, not block by block
        assert stats["blocks"] == n and stats["batched"] >= 2 and stats["served"] >= 2 * n, stats
