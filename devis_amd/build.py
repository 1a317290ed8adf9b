"""Builds devis_amd/libmsda_hip.so (the C-ABI HIP library, include/msda.h) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting .so
sits IN-TREE next to this file (git-ignored, but shipped to the GPU box with the repo snapshot).

    python -m devis_amd.build [--force] [-D...]

The library is several translation units (``csrc/*.hip``, one per kernel family, sharing ``csrc/*.h``):
they are compiled in parallel into ``devis_amd/_build/*.o`` (each with a content-hash sidecar, so an
edit recompiles only the unit it touches) and linked into one shared object.

``MSDA_LIB=/path/to/other.so`` together with ``MSDA_ENABLE_HOOKS=1`` makes :func:`lib_path` (and hence ``_native.load``) use
that file as is -- for same-box A/B runs of an experimental build -- without touching the in-tree library.  Without
``MSDA_ENABLE_HOOKS=1`` the variable is an error: a production process cannot be pointed at another build by a stray variable.
"""
import concurrent.futures
import fcntl
import glob
import hashlib
import os
import shutil
import subprocess
import sys
import warnings

# realpath, not abspath: the package may be reached through a symlink (INTEGRATION.md path A links it as
# DeVIS/src/models/ops); sources, header and the built library are located from where the files really are
HERE = os.path.dirname(os.path.realpath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libmsda_hip.so")
HASH = os.path.join(HERE, "libmsda_hip.srchash")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-munsafe-fp-atomics",          # float atomicAdd -> global_atomic_add_f32/_f64 (no CAS loop)
    "-ffp-contract=off",            # x*W-0.5 must stay a rounded product then a subtraction (which pixel cell a
                                    # point falls in); every FMA the kernels want is an explicit fmaf()
    "-Wno-pass-failed",
]


def include_dir():
    """Directory of msda.h: <repo>/include (the canonical copy, next to the package) or, for a relocated package,
    a copy shipped inside it (devis_amd/include)."""
    for d in (os.path.join(ROOT, "include"), os.path.join(HERE, "include")):
        if os.path.exists(os.path.join(d, "msda.h")):
            return d
    return None


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    inc = include_dir()
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc"))) + \
        ([os.path.join(inc, "msda.h")] if inc else [])


def _lib_override():
    """$MSDA_LIB, honoured only beside MSDA_ENABLE_HOOKS=1 (measurement / A-B runs); set without it: an error, not a silent
    switch of library (nor a silently ignored wish)."""
    path = os.environ.get("MSDA_LIB")
    if not path:
        return None
    if os.environ.get("MSDA_ENABLE_HOOKS") != "1":
        raise RuntimeError("MSDA_LIB=%s is set without MSDA_ENABLE_HOOKS=1: another build of the library is only loaded for "
                           "measurement runs (unset MSDA_LIB, or set MSDA_ENABLE_HOOKS=1)" % path)
    return path


def lib_path():
    return _lib_override() or LIB


def _digest(paths, extra=()):
    h = hashlib.sha256()
    for path in paths:
        h.update(os.path.basename(path).encode())
        if os.path.exists(path):        # (a missing file changes the hash instead of raising: is_stale() must not throw)
            with open(path, "rb") as f:
                h.update(f.read())
    h.update(" ".join(list(HIPCC_FLAGS) + list(extra)).encode())
    return h.hexdigest()


def _source_hash(extra=()):
    return _digest(sources() + _headers(), extra)


def is_stale():
    """True when the library is missing or was built from other sources / flags than the ones in the tree.
    Compared by content hash (a sidecar file written at build time), not by mtime: the repository snapshot that
    travels to the GPU box does not promise to keep timestamps.  Never raises."""
    try:
        if not os.path.exists(LIB) or not os.path.exists(HASH):
            return True
        with open(HASH) as f:
            return f.read().strip() != _source_hash()
    except OSError:
        return True


def have_compiler():
    return bool(shutil.which("hipcc")) or os.path.exists("/opt/rocm/bin/hipcc")


def _write_atomic(path, text):
    tmp = "%s.tmp.%d" % (path, os.getpid())
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def _compile_unit(hipcc, src, inc, defines, verbose):
    """One translation unit -> devis_amd/_build/<name>.o, skipped when its hash sidecar is current."""
    obj = os.path.join(OBJ, os.path.splitext(os.path.basename(src))[0] + ".o")
    want = _digest([src] + _headers(), defines)
    try:
        with open(obj + ".hash") as f:
            if f.read().strip() == want and os.path.exists(obj):
                return obj
    except OSError:
        pass
    cmd = [hipcc] + HIPCC_FLAGS + list(defines) + ["-I", inc, "-I", CSRC, "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    _write_atomic(obj + ".hash", want + "\n")
    return obj


def build(force=False, verbose=False, defines=(), out=None, jobs=None):
    """Compile the library if it is missing or was built from other sources.  Returns its path.
    ``defines`` / ``out``: an experimental build (extra -D flags) written somewhere else (see MSDA_LIB)."""
    target = out or LIB
    official = out is None and not defines
    if official and not force and not is_stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build %s" % target)
    inc = include_dir()
    if inc is None:
        raise RuntimeError("msda.h not found (looked in %s/include and %s/include)" % (ROOT, HERE))
    os.makedirs(OBJ, exist_ok=True)
    # one build at a time per tree: every rank of a multi-process launch would otherwise find the library stale
    # and compile it concurrently; the losers re-check under the lock and find it fresh
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if official and not force and not is_stale():
            return LIB
        srcs = sources()
        if force:
            for f in glob.glob(os.path.join(OBJ, "*.hash")):
                os.remove(f)
        workers = jobs or min(len(srcs), os.cpu_count() or 1, 8)
        if defines:     # experimental objects must not pose as the official ones
            objdir = os.path.join(OBJ, "exp_" + hashlib.sha256(" ".join(defines).encode()).hexdigest()[:10])
            os.makedirs(objdir, exist_ok=True)
        with concurrent.futures.ThreadPoolExecutor(max_workers=max(workers, 1)) as pool:
            if defines:
                def unit(src):
                    obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o")
                    cmd = [hipcc] + HIPCC_FLAGS + list(defines) + ["-I", inc, "-I", CSRC, "-c", src, "-o", obj]
                    if verbose:
                        print(" ".join(cmd), flush=True)
                    subprocess.check_call(cmd)
                    return obj
                objs = list(pool.map(unit, srcs))
            else:
                objs = list(pool.map(lambda s: _compile_unit(hipcc, s, inc, (), verbose), srcs))
        tmp = target + ".tmp.%d" % os.getpid()
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(tmp, target)
        if official:
            _write_atomic(HASH, _source_hash() + "\n")
        if defines:
            # the objects of an experimental build are never reused (every unit is recompiled): removed once linked -- round 4
            # left 128 MB of them in the tree, and the tree is what travels to the GPU box
            shutil.rmtree(objdir, ignore_errors=True)
    return target


def ensure():
    """What _native.load() calls: build when missing or stale (and a compiler exists); a stale library on a box
    without hipcc is loaded with a warning, a missing one raises."""
    other = _lib_override()
    if other:
        if not os.path.exists(other):
            raise RuntimeError("MSDA_LIB=%s does not exist" % other)
        return other
    if not os.path.exists(LIB) or is_stale():
        if have_compiler():
            build()
        elif os.path.exists(LIB):
            warnings.warn("devis_amd: %s was built from other sources than the ones in the tree and there is no hipcc "
                          "to rebuild it; loading it as is" % LIB, RuntimeWarning)
        else:
            raise RuntimeError("hipcc not found and no prebuilt library")
    return LIB


if __name__ == "__main__":
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")]
    print(build(force="--force" in sys.argv, verbose=True, defines=defs, out=outs[0] if outs else None))
