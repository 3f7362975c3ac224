"""Build the CPU emulation of the HIP kernels (TEST INFRASTRUCTURE ONLY)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "libspyemu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def sources():
    root = os.path.join(HERE, "..", "..")
    deps = [os.path.join(HERE, "hip_emu.h"), os.path.join(HERE, "emu_kernels.cpp"),
            os.path.join(root, "include", "spyhip.h")]
    csrc = os.path.join(root, "syncopy_amd", "csrc")
    deps += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    return deps


def build(force=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT):
        if all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in sources()):
            return OUT
    cxx = CLANG if os.path.exists(CLANG) else "g++"
    cmd = [cxx, "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-x", "c++",
           os.path.join(HERE, "emu_kernels.cpp"), "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
