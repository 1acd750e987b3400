"""The native RCCL entry of the C ABI (vg_comm_*): one-rank communicator on any GPU box, N ranks over xGMI when the box
has N >= 2 devices (skipped otherwise -- the multi-rank control flow itself is covered on CPU by
tests/test_distributed_gloo.py and on one GPU by test_gpu_solve.py::test_two_shards_with_allreduce_equal_one_problem)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(vg, model, n, cfg=2):
    from visgeom_amd import synthetic as S

    d = S.make_mono(model, n, cfg)
    p = vg.CalibrationProblem(0)
    cam = p.add_camera(model, d["init_intrinsics"])
    seq = p.add_transform(False, d["init_poses"])
    ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"])
    p.finalize()
    return p, ds, d


def test_one_rank_communicator_is_the_identity_and_the_solver_accepts_it():
    import torch

    import visgeom_amd as vg
    from visgeom_amd import distributed as D

    comm = D.Comm(D.Comm.unique_id(), 1, 0, 0)
    assert (comm.n_ranks, comm.rank) == (1, 0)
    t = torch.arange(1000, dtype=torch.float64, device="cuda")
    comm.allreduce_sum(t)
    torch.cuda.synchronize()
    assert torch.equal(t, torch.arange(1000, dtype=torch.float64, device="cuda"))
    for model in ("eucm", "mei"):
        p, _, d = _problem(vg, model, 24)
        s0 = p.solve(max_num_iterations=60)
        x0 = p.get_parameters()
        q, _, _ = _problem(vg, model, 24)
        s1 = q.solve(comm=comm, max_num_iterations=60)
        x1 = q.get_parameters()
        assert np.array_equal(x0, x1) and s0["num_iterations"] == s1["num_iterations"]
        p.close()
        q.close()
    comm.close()


def test_comm_and_host_callback_are_exclusive():
    import visgeom_amd as vg
    from visgeom_amd import capi, distributed as D

    comm = D.Comm(D.Comm.unique_id(), 1, 0, 0)
    p, _, _ = _problem(vg, "ucm", 4)
    # a one-rank communicator next to a host callback is still "one GPU through RCCL": allowed
    p.solve(comm=comm, allreduce=lambda buf: buf, max_num_iterations=3)
    p.close()
    comm.close()
    with pytest.raises(capi.VisgeomError):
        D.Comm(b"\0" * 128, 2, 5, 0)  # rank outside [0, n_ranks)


def test_n_ranks_over_rccl_equal_the_single_gpu_solve():
    import torch

    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs at least two GPUs (this box has %d)" % n)
    world = 2 if n < 4 else 4
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "tests", "rccl_worker.py")]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0 and "RCCL_WORKER_OK world=%d" % world in text, text[-4000:]
