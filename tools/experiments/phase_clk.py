"""Where the wavefronts of the tabulation kernel spend their time, by part of the kernel (shader-clock ticks summed over all
wavefronts; an instrumented build):
    tools/mkvariant.sh phclk -DMDX_WAVE_CLK -DMDX_PHASE_CLK
    gpurun -- 'MDX_LIB=tools/bin/libmdx_phclk.so python tools/experiments/phase_clk.py 10000000 "config 3|plain paired|ins 8%|del 8%"'
(tools/split_cost.py VARIANTS; the packed kernel — the batch is uploaded with its 4-bit SEQ column)."""
import ctypes, os, sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
os.environ["MDX_DBG_CLK"] = "1"
from mapdamage_amd import engine, synth
from tools.split_cost import VARIANTS
if os.environ.get('MDX_LIB'):
    engine._lib = engine.load_library(os.environ['MDX_LIB'])
NAMES = ["other (tile hand-out, rounds)", "phase 1 of a tile", "complete runs of the tile", "partial runs of the tile", "drain (events)",
         "planes -> LDS", "general pass", "lists: complete / partial", "lists: insertion runs", "lists: deletion runs",
         "final drain + fold", "prefetch area: wait + read (tile head)", "general: walks (phase 2b)", "general: compositions behind deletions", "lists: the runs themselves (of the three list rows)", "wait for the next tile's columns, its second round trip out"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
which = (sys.argv[2] if len(sys.argv) > 2 else "config 3").split("|")
ref = synth.make_genome()
batches = {w: synth.parallel_batch(dict(dict(read_len=100, paired=True, contigs=[0, 1]), **dict(VARIANTS)[w]), ref, n, 3, workers=64) for w in which}
lib = engine._lib
HAVE = hasattr(lib, 'mdx_dbg_phase_read') and os.environ.get('MDX_LIB')
with engine.DamageEngine([("s", "l")], 70, 10, 0, lgd_max=4096) as eng:
    eng.set_reference(ref)
    for w in which:
        db = eng.upload(batches[w], packed=True)
        eng.tabulate(db)
        eng.sync()
        out = np.zeros(16, np.uint64)
        if HAVE: lib.mdx_dbg_phase_read(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(1))
        eng.timing(True)
        K = 10
        for _ in range(K):
            eng.tabulate(db)
        eng.sync()
        n_launch, ms = eng.timing_read()
        eng.timing(False)
        if HAVE: lib.mdx_dbg_phase_read(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(1))
        out[0] += 1
        t = out[:16].astype(np.float64) / K
        print("== %s: %d records, kernel %.4f ms (instrumented), ticks per record %.1f" % (w, n, ms / n_launch, t[:14].sum() / n))
        for name, v in zip(NAMES, t):
            print("   %-34s %6.2f %%   %7.2f ticks per record" % (name, 100.0 * v / t[:14].sum(), v / n))
        db.free()
