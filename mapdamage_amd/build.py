"""Builds libmdx.so (HIP kernels + C-ABI) for gfx950 in-tree with hipcc."""

import os
import pathlib
import shutil
import subprocess

HERE = pathlib.Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libmdx.so"
SOURCES = ["mdx_kernels.hip", "mdx_capi.cpp", "mdx_bamio.cpp", "mdx_gbam.hip", "mdx_libsort.hip", "mdx_fasta.hip"]
HEADERS = [CSRC / "mdx_internal.h", CSRC / "mdx_inflate.h", CSRC / "mdx_deflate.h", CSRC / "mdx_crc32.h", HERE.parent / "include" / "mdx.h"]


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
    """Every source to an object of its own (side by side; an object newer than its source and the headers is kept), then
    one link.  ``extra_flags`` (A/B builds: -D...) compile everything afresh into a directory of their own."""
    if not force and not extra_flags and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wall", "-Wno-unused-function"] + list(extra_flags)
    objdir = HERE / "build" / ("obj" if not extra_flags else "obj_" + "_".join(f.strip("-").replace("=", "_") for f in extra_flags))
    objdir.mkdir(parents=True, exist_ok=True)
    newest_header = max(p.stat().st_mtime for p in HEADERS)

    def compile_one(name):
        src, obj = CSRC / name, objdir / (name + ".o")
        if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, newest_header):
            return obj
        cmd = [hipcc()] + flags + ["-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(len(SOURCES)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    # linked under another name and moved into place: a process that finds libmdx.so finds a complete one
    tmp = LIB.with_name("libmdx.so.%d.tmp" % os.getpid())
    cmd = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + [str(o) for o in objs] + ["-lz", "-lpthread", "-ldl", "-o", str(tmp)]
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
