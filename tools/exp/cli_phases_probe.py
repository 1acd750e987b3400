"""Where the wall clock of `calib a.json` goes OUTSIDE the library: the stages of the program on the parent's monotonic clock
(VG_CALIB_CLOCK_T0), from just before the process is started to the parent's wait() returning.
usage: python tools/exp/cli_phases_probe.py [images] [runs]"""
import os, subprocess, sys, tempfile, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
from visgeom_amd import benchlib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d = tempfile.mkdtemp(prefix="vg_cli_")
path, _ = benchlib.write_calib_workload(d, "mono_eucm", n)
exe = os.path.join(root, "visgeom_amd", "bin", "calib")
extra = sys.argv[3:]
for i in range(runs):
    env = dict(os.environ)
    t0 = time.clock_gettime(time.CLOCK_MONOTONIC)
    env["VG_CALIB_CLOCK_T0"] = repr(t0)
    for kv in extra:
        k, v = kv.split("=", 1); env[k] = v
    r = subprocess.run([exe, path], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, timeout=600)
    t1 = time.clock_gettime(time.CLOCK_MONOTONIC)
    st = [l.split() for l in r.stderr.decode().splitlines() if l.startswith("calib clock:")]
    for l in r.stderr.decode().splitlines():
        if l.startswith("calib phases:"): print("   ", l)
    print("run %d rc=%d  " % (i, r.returncode) + "  ".join("%s %.3f" % (x[2], float(x[3])) for x in st) + "  parent_wait %.3f" % (t1 - t0))
