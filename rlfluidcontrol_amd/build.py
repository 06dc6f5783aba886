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
# translation units, compiled in parallel: the C ABI + small kernels, the three instantiations of the step kernels
# (shipped / diagnostics compiled in / perf mode / few envs per CU: step_kernels.h), the host geometry
SOURCES = ["afc_env.hip", "step_prod.hip", "step_diag.hip", "step_perf.hip", "step_lat.hip", "geometry.cpp"]

def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".cpp"))]
    d.append(os.path.join(HERE, "..", "include", "afc_env.h"))
    return d

# -fno-slp-vectorize: left to itself the SLP vectoriser pairs scalar float operations into v_pk_*_f32 and pays for it
# in register moves (every phase here is bound by VALU issue); packed arithmetic is written out where it pays
# (the QUICK faces of advdif_v2).  Measured +2 % on the whole step.
FLAGS = (os.environ.get("AFC_EXTRA_FLAGS", "").split()) + ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


STAMP = os.path.join(HERE, "build", "flags.stamp")


def _flag_string():
    """the effective compile + link flags: a library built with other flags (AFC_EXTRA_FLAGS) is not reused silently"""
    return " ".join(FLAGS) + " | HIPCC=" + os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if any(os.path.getmtime(d) > t for d in _deps()):
        return True
    if os.environ.get("AFC_EXTRA_FLAGS") or os.path.exists(STAMP):
        # a stamp exists only next to a library this module built: compare the flags it was built with
        try:
            return open(STAMP).read() != _flag_string()
        except OSError:
            return True
    return False


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    import fcntl
    # several ranks of one node may get here together (torchrun): serialise, re-check, build to a
    # temporary name and rename so that nobody ever loads a half-written library
    with open(os.path.join(HERE, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if force or needs_build():
                hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
                tmp = LIB + ".tmp.%d" % os.getpid()
                objdir = os.path.join(HERE, "build")
                os.makedirs(objdir, exist_ok=True)
                jobs = []
                try:
                    for src in SOURCES:
                        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".%d.o" % os.getpid())
                        cmd = [hipcc] + FLAGS + ["-c", "-o", obj, os.path.join(CSRC, src)]
                        if verbose:
                            print(" ".join(cmd))
                        jobs.append((obj, cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
                    failed = []
                    for obj, cmd, pr in jobs:
                        out = pr.communicate()[0].decode(errors="replace")
                        if pr.returncode != 0:
                            failed.append((cmd, out))
                        elif verbose and out.strip():
                            print(out)
                    if failed:
                        for cmd, out in failed:
                            sys.stderr.write(" ".join(cmd) + "\n" + out + "\n")
                        raise subprocess.CalledProcessError(1, failed[0][0])
                    # the extra flags reach the link step too (a CPU-side -fsanitize needs its runtime there)
                    extra = os.environ.get("AFC_EXTRA_FLAGS", "").split()
                    subprocess.check_call([hipcc] + extra + ["-shared", "-fPIC", "--offload-arch=gfx950", "-o", tmp] +
                                          [j[0] for j in jobs])
                    os.replace(tmp, LIB)
                    with open(STAMP, "w") as f:
                        f.write(_flag_string())
                finally:  # object files never outlive the call, whether a translation unit failed or not
                    for obj, _, pr in jobs:
                        if pr.poll() is None:
                            pr.wait()
                        if os.path.exists(obj):
                            os.remove(obj)
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
