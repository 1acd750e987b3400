"""Multi-GPU plumbing: one process per GPU, images sharded over ranks, global parameters replicated.

The path has exactly one exchange step: the sum of the small normal-equation blocks (global block U, gradient,
cost, Schur contribution) -- a few KB per iteration, latency bound, so the collective is a plain summing
all-reduce (RCCL through torch.distributed's "nccl" backend on GPUs, "gloo" in the CPU tests).  Per-image data
never leaves its rank.
"""
import numpy as np


def shard_range(n_items, rank, world_size):
    """contiguous, balanced image range [lo, hi) of `rank` (the first n_items % world_size ranks get one more)."""
    base, extra = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def make_allreduce(group=None, device=None):
    """Returns allreduce(buf: np.ndarray float64) summing `buf` over the ranks of `group`, in place -- the callback
    CalibrationProblem.solve(allreduce=...) expects.  With the nccl backend the buffer is staged through a device
    tensor (RCCL reduces device memory); with gloo it is reduced in host memory."""
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)
    if backend == "nccl" and device is None:
        device = torch.device("cuda", torch.cuda.current_device())

    def allreduce(buf):
        t = torch.from_numpy(buf)
        if device is not None:
            g = t.to(device)
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
            t.copy_(g.cpu())
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return buf

    return allreduce


def pack_normal_blocks(U, g, cost2, extra=()):
    """the message of one iteration: [U (G*G) | g (G) | cost2 | extra...] as one float64 buffer."""
    return np.concatenate([np.asarray(U, float).ravel(), np.asarray(g, float).ravel(), [float(cost2)],
                           np.asarray(extra, float).ravel()])


def unpack_normal_blocks(buf, G, n_extra=0):
    U = buf[:G * G].reshape(G, G)
    g = buf[G * G:G * G + G]
    cost2 = float(buf[G * G + G])
    return U, g, cost2, buf[G * G + G + 1:G * G + G + 1 + n_extra]
