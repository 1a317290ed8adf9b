"""Builds devis_amd/libmsda_hip.so (the C-ABI HIP library, include/msda.h) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting .so
sits IN-TREE next to this file (git-ignored, but shipped to the GPU box with the repo snapshot).

    python -m devis_amd.build [--force]
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "msda_hip.hip")
INC = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "libmsda_hip.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-munsafe-fp-atomics",          # float atomicAdd -> global_atomic_add_f32/_f64 (no CAS loop)
    "-ffp-contract=off",            # x*W-0.5 must stay a rounded product then a subtraction (which pixel cell a
                                    # point falls in); every FMA the kernels want is an explicit fmaf()
    "-Wno-pass-failed",
]


def lib_path():
    return LIB


def is_stale():
    if not os.path.exists(LIB):
        return True
    deps = [SRC, os.path.join(INC, "msda.h"), os.path.abspath(__file__)]
    return os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile the library if it is missing or older than its sources.  Returns its path."""
    if not force and not is_stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build %s" % LIB)
    tmp = LIB + ".tmp.%d" % os.getpid()
    cmd = [hipcc] + HIPCC_FLAGS + ["-I", INC, SRC, "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
