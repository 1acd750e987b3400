"""Multi-GPU plumbing: one process per GPU, images sharded over ranks, global parameters replicated.

The path has exactly one exchange step: the sum of the small normal-equation blocks (global block U, gradient,
cost, Schur contribution) -- a few KB per iteration, latency bound, so the collective is a plain summing
all-reduce (RCCL through torch.distributed's "nccl" backend on GPUs, "gloo" in the CPU tests).  Per-image data
never leaves its rank.
"""
import ctypes

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


class Comm:
    """vg_comm: an RCCL communicator owned by the native library (include/visgeom_amd.h, "multi-GPU").  The solver and
    the normal-equation build reduce their DEVICE buffers through it, in place, on the problem's stream -- no host
    staging, no torch tensor in between.  torch.distributed is only the courier of the 128-byte unique id."""

    def __init__(self, id_bytes, n_ranks, rank, device):
        from . import capi

        self._lib = capi.load()
        h = ctypes.c_void_p()
        capi.check(self._lib.vg_comm_create(ctypes.byref(h), bytes(id_bytes), int(n_ranks), int(rank), int(device)))
        self._h = h
        self.n_ranks, self.rank, self.device = int(n_ranks), int(rank), int(device)

    @classmethod
    def replicated(cls, replicas, device=0):
        """a communicator standing for `replicas` ranks with identical shards (vg_comm_create_replicated): no RCCL"""
        from . import capi

        self = cls.__new__(cls)
        self._lib = capi.load()
        h = ctypes.c_void_p()
        capi.check(self._lib.vg_comm_create_replicated(ctypes.byref(h), int(replicas), int(device)))
        self._h = h
        self.n_ranks, self.rank, self.device = int(replicas), 0, int(device)
        return self

    @classmethod
    def local_group(cls, n_ranks, device=0):
        """n_ranks communicators for n_ranks THREADS of this process on one device (vg_comm_create_local): the solver's
        real multi-rank data flow with different shards per rank, on a one-GPU box"""
        from . import capi

        lib = capi.load()
        hs = (ctypes.c_void_p * int(n_ranks))()
        capi.check(lib.vg_comm_create_local(hs, int(n_ranks), int(device)))
        out = []
        for r in range(int(n_ranks)):
            self = cls.__new__(cls)
            self._lib = lib
            self._h = ctypes.c_void_p(hs[r])
            self.n_ranks, self.rank, self.device = int(n_ranks), r, int(device)
            out.append(self)
        return out

    @staticmethod
    def unique_id():
        from . import capi

        buf = ctypes.create_string_buffer(128)
        capi.check(capi.load().vg_comm_unique_id(buf))
        return buf.raw

    @property
    def handle(self):
        return self._h

    def allreduce_sum(self, tensor, stream=None):
        """in-place sum of a float64 CUDA tensor over all ranks, enqueued on `stream` (default: torch's current one)"""
        import torch

        from . import capi

        assert tensor.is_cuda and tensor.dtype == torch.float64 and tensor.is_contiguous()
        if stream is None:
            stream = torch.cuda.current_stream(tensor.device).cuda_stream
        capi.check(self._lib.vg_comm_allreduce_sum(self._h, ctypes.c_void_p(tensor.data_ptr()), tensor.numel(),
                                                   ctypes.c_void_p(stream)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vg_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_comm(device, group=None):
    """One native RCCL communicator spanning the ranks of a torch.distributed group (any backend): rank 0 draws the
    unique id, the group broadcasts its 128 bytes, every rank joins on its own GPU."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return Comm(box[0], world, rank, device)
