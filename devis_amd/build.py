"""Builds devis_amd/libmsda_hip.so (the C-ABI HIP library, include/msda.h) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting .so
sits IN-TREE next to this file (git-ignored, but shipped to the GPU box with the repo snapshot).

    python -m devis_amd.build [--force]
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "msda_hip.hip")
INC = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "libmsda_hip.so")
HASH = os.path.join(HERE, "libmsda_hip.srchash")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-munsafe-fp-atomics",          # float atomicAdd -> global_atomic_add_f32/_f64 (no CAS loop)
    "-ffp-contract=off",            # x*W-0.5 must stay a rounded product then a subtraction (which pixel cell a
                                    # point falls in); every FMA the kernels want is an explicit fmaf()
    "-Wno-pass-failed",
]


def lib_path():
    return LIB


def _source_hash():
    h = hashlib.sha256()
    for path in (SRC, os.path.join(INC, "msda.h")):
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def is_stale():
    """True when the library is missing or was built from other sources / flags than the ones in the tree.
    Compared by content hash (a sidecar file written at build time), not by mtime: the repository snapshot that
    travels to the GPU box does not promise to keep timestamps."""
    if not os.path.exists(LIB) or not os.path.exists(HASH):
        return True
    with open(HASH) as f:
        return f.read().strip() != _source_hash()


def have_compiler():
    return bool(shutil.which("hipcc")) or os.path.exists("/opt/rocm/bin/hipcc")


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
    with open(HASH, "w") as f:
        f.write(_source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
