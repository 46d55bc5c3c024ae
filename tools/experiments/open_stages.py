"""Where GpuBamStream's construction spends its time (mdx_gbam_open, the header as Python objects, mdx_gbam_configure).
Run on the GPU box: python tools/experiments/open_stages.py"""
import ctypes, os, sys, tempfile, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from mapdamage_amd import engine, sam, synth
ref = synth.make_genome()
batch = synth.parallel_batch("config3_batch", ref, 8_000_000, seed=3003, workers=64)
tmp = tempfile.mkdtemp(prefix="mdx_open_")
path = os.path.join(tmp, "c3.bam")
sam.write_bam(path, batch, ref.names, ref.lengths, [{"ID": "rg1", "SM": "synthetic", "LB": "lib1"}], rg_of_record=["rg1"] * batch.n, workers=64)
with open(path, "rb") as fh:
    while fh.read(1 << 26): pass
with engine.DamageEngine([("synthetic", "lib1")], 70, 10, 0) as eng:
    eng.set_reference(ref)
    lib = eng._lib
    for rep in range(4):
        g = ctypes.c_void_p()
        t0 = time.perf_counter()
        rc = lib.mdx_gbam_open(eng._ctx, path.encode(), ctypes.byref(g)); assert rc == 0
        t1 = time.perf_counter()
        hdr = sam._native_header(lib, lib.mdx_gbam_header(g))
        t2 = time.perf_counter()
        arr = (ctypes.c_char_p * 1)(b"rg1"); libs = (ctypes.c_int32 * 1)(0)
        rc = lib.mdx_gbam_configure(g, 1, arr, libs, -1, 0, 0); assert rc == 0
        t3 = time.perf_counter()
        lib.mdx_gbam_close(g)
        t4 = time.perf_counter()
        print("rep %d: mdx_gbam_open %.2f ms, header objects %.2f, configure %.2f, close (nothing decoded) %.2f" % (rep, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3)), flush=True)
