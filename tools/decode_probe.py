"""Where the BAM decode time goes (GPU box host): the C++ call alone for several thread counts, then the Python wrapper."""
import ctypes, os, sys, tempfile, time, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mapdamage_amd import sam, synth
from mapdamage_amd.engine import load_library
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ref = synth.make_genome()
batch = synth.config3_batch(ref, n, seed=3)
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "x.bam")
    t = time.perf_counter()
    sam.write_bam(path, batch, ref.names, ref.lengths, [{"ID": "rg1", "SM": "s", "LB": "l"}], rg_of_record=["rg1"] * n)
    print("write_bam", time.perf_counter() - t, "cpus", os.cpu_count())
    lib = load_library()
    for thr in (1, 4, 16, 64):
        h = ctypes.c_void_p()
        t = time.perf_counter()
        lib.mdx_bam_read(path.encode(), ctypes.c_int(thr), ctypes.byref(h))
        print("mdx_bam_read threads", thr, time.perf_counter() - t)
        lib.mdx_bam_free(h)
    t = time.perf_counter()
    al = sam.read_bam_native(path)
    print("read_bam_native (wrapper incl.)", time.perf_counter() - t)
