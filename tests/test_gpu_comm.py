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


@pytest.mark.parametrize("replicas", [2, 8])
def test_multi_rank_control_flow_on_one_gpu_with_a_replicated_communicator(replicas):
    """vg_comm_create_replicated: this rank stands for `replicas` ranks holding the same shard, every sum over ranks is
    replicas x the local value.  The solver then runs exactly what N real ranks run -- packed collectives of the summed
    Gram blocks / step scalars / Schur complement, the summable 2-norm instead of the max-norm of the pose gradient, no
    speculative queueing -- and must end at the one-rank optimum with `replicas` times the cost: mono (device-resident loop),
    stereo (several datasets, merged Gram launch) and a rig (host-driven loop, 45 global columns)."""
    import visgeom_amd as vg
    from visgeom_amd import distributed as D, synthetic as S

    comm = D.Comm.replicated(replicas)
    assert (comm.n_ranks, comm.rank) == (replicas, 0)

    def check(build, iters=150):
        p = build()
        s0 = p.solve(max_num_iterations=iters)
        x0 = p.get_parameters()
        p.close()
        q = build()
        s1 = q.solve(comm=comm, max_num_iterations=iters)
        x1 = q.get_parameters()
        q.close()
        assert s0["termination"].startswith("CONVERGENCE") and s1["termination"].startswith("CONVERGENCE"), (s0, s1)
        assert abs(s1["final_cost"] - replicas * s0["final_cost"]) <= 1e-9 * replicas * s0["final_cost"]
        assert abs(s1["initial_cost"] - replicas * s0["initial_cost"]) <= 1e-12 * replicas * s0["initial_cost"]
        scale = np.maximum(np.abs(x0), 1.0)
        assert np.max(np.abs(x1 - x0) / scale) < 1e-6, np.max(np.abs(x1 - x0) / scale)   # both stop at rounding level

    for model in ("eucm", "mei"):
        check(lambda: _problem(vg, model, 40)[0])

    st = S.make_stereo(30, sigma=0.1)

    def stereo():
        p = vg.CalibrationProblem(0)
        c1 = p.add_camera("eucm", st["init_intrinsics1"])
        c2 = p.add_camera("eucm", st["init_intrinsics2"])
        x12 = p.add_transform(True, st["init_xi12"])
        seq = p.add_transform(False, st["init_poses"])
        p.add_dataset(c1, [(seq, 0)], st["board"], st["corners1"])
        p.add_dataset(c2, [(x12, 1), (seq, 0)], st["board"], st["corners2"])
        p.finalize()
        return p

    check(stereo)

    r = S.make_rig(25, sigma=0.1)

    def rig():
        p = vg.CalibrationProblem(0)
        cams = [p.add_camera(m, r["init_intrinsics"][k]) for k, m in enumerate(r["models"])]
        x1k = [p.add_transform(True, r["init_xi1k"][k]) for k in range(3)]
        seq = p.add_transform(False, r["init_poses"])
        p.add_dataset(cams[0], [(seq, 0)], r["board"], r["corners"][0])
        for k in range(3):
            p.add_dataset(cams[k + 1], [(x1k[k], 1), (seq, 0)], r["board"], r["corners"][k + 1])
        p.finalize()
        return p

    check(rig, iters=300)
    comm.close()
