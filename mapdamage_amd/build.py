"""Builds libmdx.so (HIP kernels + C-ABI) for gfx950 in-tree with hipcc."""

import os
import pathlib
import shutil
import subprocess

HERE = pathlib.Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libmdx.so"
SOURCES = ["mdx_kernels.hip", "mdx_capi.cpp", "mdx_bamio.cpp", "mdx_gbam.hip", "mdx_libsort.hip"]
HEADERS = [CSRC / "mdx_internal.h", CSRC / "mdx_inflate.h", CSRC / "mdx_crc32.h", HERE.parent / "include" / "mdx.h"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP extension cannot be built")
    return exe


def needs_build():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + HEADERS
    return any(p.stat().st_mtime > t for p in deps)


def build_lib(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-x", "hip", "-Wall", "-Wno-unused-function"]
    cmd += list(extra_flags)
    cmd += [str(CSRC / s) for s in SOURCES]
    # linked under another name and moved into place: a process that finds libmdx.so finds a complete one
    tmp = LIB.with_name("libmdx.so.%d.tmp" % os.getpid())
    cmd += ["-lz", "-lpthread", "-ldl", "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    finally:
        if tmp.exists():
            tmp.unlink()
    return LIB


def build_lib_locked():
    """build_lib() under an exclusive file lock: of several processes started together (one rank per GPU) one compiles,
    the others wait for it and find the library built."""
    import fcntl
    # (the check is made under the lock: a rank that arrives while another one links must wait for it, not load half a file)
    with open(HERE / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return build_lib()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    import sys
    build_lib(force=True, verbose=True, extra_flags=sys.argv[1:])
    print("built", LIB)
