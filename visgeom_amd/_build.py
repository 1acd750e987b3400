"""In-tree build of the HIP library (gfx950 only).  `python -m visgeom_amd._build` or
`__graft_entry__.build()`.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libvisgeom_amd.so")

# -ffp-contract=off: keep the per-corner arithmetic in the reference's evaluation order (no FMA
# fusion), so that GPU and CPU checker differ only through libm-vs-ocml trig in the chain prep.  The emit
# kernel is HBM bound, the extra VALU instructions are not on the critical path.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# the A/B and test switches behind vg_debug_set(); VG_PRODUCTION=1 builds the library without them
if not os.environ.get("VG_PRODUCTION"):
    HIPCC_FLAGS.append("-DVG_DEBUG_HOOKS")


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]


def _deps():
    out = []
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            if f.endswith((".hip", ".hpp", ".h", ".cpp")):
                out.append(os.path.join(d, f))
    return out


STAMP = os.path.join(LIB_DIR, "libvisgeom_amd.sources.sha256")


def sources_digest():
    """sha256 over the compiler flags and the CONTENT of every source / header the library is built from (a snapshot copied
    to another machine has fresh mtimes everywhere: modification times say nothing about what a library was built from)"""
    import hashlib

    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for f in sorted(_deps()):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def up_to_date():
    if not os.path.exists(LIB) or not os.path.exists(os.path.join(PKG, "bin", "calib")) or not os.path.exists(STAMP):
        return False
    with open(STAMP) as fh:
        return fh.read().strip() == sources_digest()


def build(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc] + HIPCC_FLAGS + ["-I", os.path.join(ROOT, "include"), "-o", LIB] + sources()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    build_cli(verbose)
    with open(STAMP, "w") as fh:
        fh.write(sources_digest() + "\n")
    return LIB


BIN_DIR = os.path.join(PKG, "bin")
CLI = os.path.join(BIN_DIR, "calib")


def build_cli(verbose=False):
    """the reference's `calib file1.json [file2.json ...]` entry point: a host-only C++ program on the C ABI"""
    os.makedirs(BIN_DIR, exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", os.path.join(CSRC, "calib_main.cpp"), "-o", CLI,
           "-L" + LIB_DIR, "-lvisgeom_amd", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
