"""Stage times of the GPU-side BAM decode (MDX_BAM_TIMING=1: allocate / upload / inflate / scan / unpack per slab) on a
synthetic config-3 BAM, against the host decoder.  Run on the GPU box: python tools/gpu_decode_timing.py [reads]"""
import os
import pathlib
import sys
import tempfile
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["MDX_BAM_TIMING"] = "1"

from mapdamage_amd import sam, synth  # noqa: E402
from mapdamage_amd.engine import DamageEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
ref = synth.make_genome()
batch = synth.config3_batch(ref, n, seed=3)
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "c3.bam")
    sam.write_bam(path, batch, ref.names, ref.lengths, [{"ID": "rg1", "SM": "synthetic", "LB": "lib1"}], rg_of_record=["rg1"] * n)
    with DamageEngine([("synthetic", "lib1")], 70, 10, 0) as eng:
        eng.set_reference(ref)
        for rep in range(3):
            eng.reset(); eng.sync()
            t0 = time.perf_counter()
            with sam.GpuBamStream(eng, path, readgroups=[("rg1", 0)], chunk_bytes=int(float(os.environ.get("MDX_SLAB_MB", "256")) * (1 << 20))) as g:
                t1 = time.perf_counter()
                while True:
                    v = g.next_view()
                    if v is None:
                        break
                    eng.tabulate_view(v)
                eng.finish()
                t2 = time.perf_counter()
            print("rep %d: open %.1f ms, slabs + tabulate %.1f ms, close %.1f ms" % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (time.perf_counter() - t2)), file=sys.stderr)
