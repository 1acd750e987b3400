"""How the CPU baseline (the oracle port, OpenMP over images) scales with threads on the GPU box's host: evals/s of the headline
pass (EUCM, 10 000 images x 96 corners, residual + all Jacobian blocks) for 1 .. all logical CPUs.  CPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import vgo
from visgeom_amd import synthetic

d = synthetic.make_mono("eucm", 10000, 1)
n, N, K = d["corners"].shape[0], d["board"].shape[0], 6
pv = np.concatenate([d["init_intrinsics"], d["init_poses"].ravel()])
out = (np.empty((n, 2 * N)), np.empty((n, 2 * N, K)), [np.empty((n, 2 * N, 6))])
seq = np.arange(n)
print("logical CPUs", os.cpu_count(), "allowed", len(os.sched_getaffinity(0)), "omp max", vgo.max_threads())
for th in (1, 2, 4, 8, 16, 32, 64, 128, 192, 256):
    if th > (os.cpu_count() or 1):
        break
    vgo.eval_dataset(0, [0], d["board"], d["corners"], pv, 0, [K], [6], seq, threads=th, out=out)
    t, r = time.perf_counter(), 0
    while time.perf_counter() - t < 2.0:
        vgo.eval_dataset(0, [0], d["board"], d["corners"], pv, 0, [K], [6], seq, threads=th, out=out)
        r += 1
    el = time.perf_counter() - t
    # cost-only passes (no 200 MB of rows to write): tells arithmetic scaling from memory scaling
    t2, r2 = time.perf_counter(), 0
    while time.perf_counter() - t2 < 1.0:
        vgo.eval_dataset(0, [0], d["board"], d["corners"], pv, 0, [K], [6], seq, want_jac=False, threads=th, out=None)
        r2 += 1
    el2 = time.perf_counter() - t2
    print("threads %3d: %.3e evals/s with Jacobians, %.3e cost only" % (th, r * n * N / el, r2 * n * N / el2))
