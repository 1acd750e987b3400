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
               "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-inline-asm", "-Wno-constant-logical-operand"]
# the A/B and test switches behind vg_debug_set(): the library tests/ and bench.py drive.  build_production() makes the
# library that ships -- same sources without this switch (no hook table, every switch its default at compile time,
# vg_debug_set not exported) -> lib/production/libvisgeom_amd.so
HOOKS_FLAG = "-DVG_DEBUG_HOOKS"
HIPCC_FLAGS.append(HOOKS_FLAG)
PRODUCTION_LIB = os.path.join(LIB_DIR, "production", "libvisgeom_amd.so")


def sources():
    """the translation units of the library: every .hip file of csrc/ (vg_capi: problem assembly + emit; vg_gram_tu: normal
    equations; vg_solver_tu: LM / Schur + communicator; vg_refine_tu: per-image pose LM; vg_frontend_tu: calibration JSON;
    vg_local_tu: localization costs)"""
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]


OBJ_DIR = os.path.join(LIB_DIR, "obj")
STAMP = os.path.join(LIB_DIR, "libvisgeom_amd.sources.sha256")
_INCLUDE = None


def _closure(path, seen=None):
    """the file and every project header it includes, transitively (quoted includes, resolved against csrc/ and include/)"""
    import re

    global _INCLUDE
    if _INCLUDE is None:
        _INCLUDE = re.compile(r'^\s*#\s*include\s+["<]([^">]+)[">]', re.M)
    seen = seen if seen is not None else set()
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path) as fh:
        text = fh.read()
    for inc in _INCLUDE.findall(text):
        for base in (os.path.dirname(path), CSRC, os.path.join(ROOT, "include")):
            cand = os.path.normpath(os.path.join(base, inc))
            if os.path.exists(cand):
                _closure(cand, seen)
                break
    return seen


_EXTRA_FLAGS = []   # build_variant(): additional -D switches of an A/B library (tools/exp)


def unit_digest(src):
    """sha256 over the compiler flags and the CONTENT of a translation unit and of every header it reaches (a snapshot copied
    to another machine has fresh mtimes everywhere: modification times say nothing about what an object was built from)"""
    import hashlib

    h = hashlib.sha256(" ".join(HIPCC_FLAGS + _EXTRA_FLAGS).encode())
    for f in sorted(_closure(src)):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def sources_digest():
    import hashlib

    h = hashlib.sha256()
    for src in sources():
        h.update(unit_digest(src).encode())
    with open(os.path.join(CSRC, "calib_main.cpp"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()


def up_to_date():
    if not os.path.exists(LIB) or not os.path.exists(os.path.join(PKG, "bin", "calib")) or not os.path.exists(STAMP):
        return False
    with open(STAMP) as fh:
        return fh.read().strip() == sources_digest()


def _compile_unit(hipcc, src, verbose, obj_dir=None):
    """one translation unit -> lib/obj/<name>.o, skipped when the object was built from the same content"""
    name = os.path.splitext(os.path.basename(src))[0]
    obj_dir = obj_dir or OBJ_DIR
    obj, stamp = os.path.join(obj_dir, name + ".o"), os.path.join(obj_dir, name + ".sha256")
    digest = unit_digest(src)
    if os.path.exists(obj) and os.path.exists(stamp):
        with open(stamp) as fh:
            if fh.read().strip() == digest:
                return obj, 0.0
    import time

    t0 = time.time()
    cmd = [hipcc] + [f for f in HIPCC_FLAGS + _EXTRA_FLAGS if f != "-shared"] + ["-I", os.path.join(ROOT, "include"), "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(stamp, "w") as fh:
        fh.write(digest + "\n")
    return obj, time.time() - t0


def build(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    import time
    from concurrent.futures import ThreadPoolExecutor

    t0 = time.time()
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:   # the units compile side by side
        done = list(ex.map(lambda s: _compile_unit(hipcc, s, verbose), srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [o for o, _ in done]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    if verbose:
        print("built %d translation units in %.1f s wall (%s)" % (len(srcs), time.time() - t0, ", ".join(
            "%s %.1f s" % (os.path.basename(o), t) for o, t in done)), file=sys.stderr)
    build_cli(verbose)
    with open(STAMP, "w") as fh:
        fh.write(sources_digest() + "\n")
    return LIB


def build_production(verbose=False):
    """The library that ships: the same translation units without VG_DEBUG_HOOKS.  Objects are cached by content like the
    default build's (lib/obj_production/), so an unchanged tree costs nothing."""
    return build_variant("production", [], verbose=verbose, out=PRODUCTION_LIB, drop_flags=[HOOKS_FLAG])


def build_variant(name, extra_flags, verbose=False, out=None, drop_flags=()):
    """An A/B library for tools/exp probes: the same sources with additional -D switches -> lib/variants/libvisgeom_amd_<name>.so
    (objects under lib/obj_<name>/; git-ignored like the product library, and like it carried to the GPU box by gpurun)."""
    global _EXTRA_FLAGS
    from concurrent.futures import ThreadPoolExecutor

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    obj_dir = os.path.join(LIB_DIR, "obj_" + name)
    out = out or os.path.join(LIB_DIR, "variants", "libvisgeom_amd_%s.so" % name)
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    _EXTRA_FLAGS = list(extra_flags)
    dropped = [f for f in drop_flags if f in HIPCC_FLAGS]
    for f in dropped:
        HIPCC_FLAGS.remove(f)
    try:
        srcs = sources()
        with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
            done = list(ex.map(lambda s: _compile_unit(hipcc, s, verbose, obj_dir), srcs))
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [o for o, _ in done])
    finally:
        _EXTRA_FLAGS = []
        HIPCC_FLAGS.extend(dropped)
    return out


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
    if "--production" in sys.argv:
        print(build_production(verbose=True))
    elif "--variant" in sys.argv:   # python -m visgeom_amd._build --variant NAME -DFLAG [-DFLAG ...]
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
