"""One rank of the RCCL test (tests/test_gpu_comm.py): launched by torch.distributed.run, one process per GPU.
Every rank holds a contiguous shard of the images and solves with the native communicator; rank 0 also solves the
whole set alone and the two answers must agree ("sharded result equals the 1-GPU solve to 1e-8")."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    from visgeom_amd import CalibrationProblem, distributed as D, synthetic as S

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    comm = D.make_comm(local)
    assert comm.n_ranks == world and comm.rank == rank
    # the collective itself: in-place sum of a device buffer
    t = torch.full((300,), float(rank + 1), dtype=torch.float64, device="cuda")
    comm.allreduce_sum(t)
    torch.cuda.synchronize()
    assert torch.all(t == world * (world + 1) / 2)

    n_total, model = 96, os.environ.get("VG_TEST_MODEL", "mei")
    lo, hi = D.shard_range(n_total, rank, world)
    d = S.make_mono(model, hi - lo, 4, first_image=lo)
    p = CalibrationProblem(local)
    cam = p.add_camera(model, d["init_intrinsics"])
    seq = p.add_transform(False, d["init_poses"])
    p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"])
    p.finalize()
    summ = p.solve(comm=comm, max_num_iterations=100)
    x = p.get_parameters()
    K = d["init_intrinsics"].size
    intr = torch.tensor(x[:K], dtype=torch.float64, device="cuda")
    ref = intr.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(intr, ref), "ranks must end with bit-identical global parameters"
    if rank == 0:
        dfull = S.make_mono(model, n_total, 4)
        q = CalibrationProblem(local)
        c2 = q.add_camera(model, dfull["init_intrinsics"])
        s2 = q.add_transform(False, dfull["init_poses"])
        q.add_dataset(c2, [(s2, 0)], dfull["board"], dfull["corners"])
        q.finalize()
        s1 = q.solve(max_num_iterations=100)
        y = q.get_parameters()
        rel = np.max(np.abs(x[:K] - y[:K]) / np.maximum(np.abs(y[:K]), 1.0))
        assert rel <= 1e-8, rel
        assert abs(summ["final_cost"] - s1["final_cost"]) <= 1e-9 * s1["final_cost"]
        # this rank's poses are the first (hi - lo) poses of the full problem
        assert np.max(np.abs(x[K:] - y[K:K + 6 * (hi - lo)])) <= 1e-7
        print("RCCL_WORKER_OK world=%d rel=%.3e iters=%d/%d" % (world, rel, summ["num_iterations"], s1["num_iterations"]))
        q.close()
    p.close()
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
