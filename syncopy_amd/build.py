"""Build libspyhip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

`python -m syncopy_amd.build` or `syncopy_amd.build.build()`.  The library is
kept next to this file (git-ignored, but it travels to the GPU box with the
repository snapshot).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libspyhip.so")
ARCH = "gfx950"


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, "..", "include", "spyhip.h"))
    return deps


def _fresh(obj):
    """True if `obj` is newer than every file its compiler-written dependency list (obj.d) names."""
    dep = obj + ".d"
    if not (os.path.exists(obj) and os.path.exists(dep)):
        return False
    t = os.path.getmtime(obj)
    try:
        with open(dep) as fh:
            names = fh.read().replace("\\\n", " ").split()
    except OSError:
        return False
    for n in names[1:]:
        if n.endswith(":"):
            continue
        if not os.path.exists(n) or os.path.getmtime(n) > t:
            return False
    return True


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libspyhip.so cannot be built")
    return exe


def build(force=False, verbose=False):
    if not force and os.path.exists(LIB) and all(
            os.path.getmtime(LIB) >= os.path.getmtime(d) for d in _deps()):
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    procs = []
    objs = []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and _fresh(obj):
            continue
        cmd = [hipcc()] + flags + ["-MD", "-MF", obj + ".d", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
