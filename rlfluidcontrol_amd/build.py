"""Builds libafc_env.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m rlfluidcontrol_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off: the reference (Java) never fuses
multiply-add, and parity with the oracle is checked at the bit level wherever the
reference's operation order is kept.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libafc_env.so")
SOURCES = ["afc_env.hip", "geometry.cpp"]

def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".cpp"))]
    d.append(os.path.join(HERE, "..", "include", "afc_env.h"))
    return d

FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
