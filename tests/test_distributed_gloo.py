"""N > 1 path on CPU: two gloo ranks (world_size 2).  There is no CPU product path, so the per-rank numbers are
produced by the oracle (allowed inside tests/); what is under test is the product's own multi-rank logic:
visgeom_amd.distributed (sharding, packing, the summing all-reduce callback used by CalibrationProblem.solve) and
synthetic's per-shard streams -- the sharded, all-reduced normal-equation blocks must equal the single-process ones."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from oracle import vgo
from visgeom_amd import distributed as D, synthetic as S

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n_total, model, K = 37, "eucm", 6
lo, hi = D.shard_range(n_total, rank, world)
d = S.make_mono(model, hi - lo, 3, first_image=lo)                  # this rank's images only
pv = np.concatenate([d["init_intrinsics"], d["init_poses"].ravel()])
r, ji, jm = vgo.eval_dataset(vgo.MODEL_EUCM, [0], d["board"], d["corners"], pv, 0, [K], [6], np.arange(hi - lo))
grams, total = vgo.dataset_gram(r, ji, jm)
U, g, cost2 = total[:K, :K], total[:K, -1], total[-1, -1]            # global block of the arrow system
buf = D.pack_normal_blocks(U, g, cost2, extra=[hi - lo])
allreduce = D.make_allreduce()
allreduce(buf)                                                        # the one collective of the path
U2, g2, c2, extra = D.unpack_normal_blocks(buf, K, 1)
np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), U=U2, g=g2, cost2=c2, n=extra, lo=lo, hi=hi)
dist.barrier()
dist.destroy_process_group()
"""


def test_two_gloo_ranks_reduce_to_the_single_process_blocks(tmp_path):
    from oracle import vgo
    from visgeom_amd import distributed as D, synthetic as S

    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)

    n_total, K = 37, 6
    d = S.make_mono("eucm", n_total, 3)
    pv = np.concatenate([d["init_intrinsics"], d["init_poses"].ravel()])
    r, ji, jm = vgo.eval_dataset(vgo.MODEL_EUCM, [0], d["board"], d["corners"], pv, 0, [K], [6], np.arange(n_total))
    _, total = vgo.dataset_gram(r, ji, jm)
    got = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(2)]
    assert (int(got[0]["lo"]), int(got[0]["hi"]), int(got[1]["lo"]), int(got[1]["hi"])) == (0, 19, 19, 37)
    for g in got:
        assert int(g["n"][0]) == n_total
        assert np.linalg.norm(g["U"] - total[:K, :K]) <= 1e-13 * np.linalg.norm(total[:K, :K])
        assert np.linalg.norm(g["g"] - total[:K, -1]) <= 1e-13 * np.linalg.norm(total[:K, -1])
        assert abs(float(g["cost2"]) - total[-1, -1]) <= 1e-13 * total[-1, -1]
    # both ranks hold bit-identical reduced blocks -> they take identical solver branches
    assert np.array_equal(got[0]["U"], got[1]["U"]) and np.array_equal(got[0]["g"], got[1]["g"])


def test_shard_range_is_a_partition():
    from visgeom_amd.distributed import shard_range

    for n in (0, 1, 7, 8, 10000, 10001):
        for w in (1, 2, 3, 4, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
