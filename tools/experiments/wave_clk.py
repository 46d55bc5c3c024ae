"""Per-wavefront start / end-of-tile-loop / end clocks of the tabulation kernel: how much of a launch is the tail of its
slowest wavefronts, and who they are ("The hand-out of tiles").  Needs the instrumented build:
    tools/mkvariant.sh clk -DMDX_WAVE_CLK
    gpurun -- 'MDX_LIB=tools/bin/libmdx_clk.so python tools/experiments/wave_clk.py 25000000 ["config 3" | "config 4" | ...]'   (tools/split_cost.py VARIANTS)
(three stores of the 100 MHz clock per wavefront behind -DMDX_WAVE_CLK; the library of the tree does not carry them;
MDX_SEQ_4BIT=1 for the packed kernel: 512 x 8 wavefronts)."""
import ctypes, os, sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
os.environ["MDX_DBG_CLK"] = "1"
from mapdamage_amd import engine, synth
from tools.split_cost import VARIANTS
if os.environ.get('MDX_LIB'):
    engine._lib = engine.load_library(os.environ['MDX_LIB'])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
kw = dict(VARIANTS)[sys.argv[2] if len(sys.argv) > 2 else "config 3"]
ref = synth.make_genome()
b = synth.parallel_batch(dict(dict(read_len=100, paired=True, contigs=[0, 1]), **kw), ref, n, 3, workers=64)
with engine.DamageEngine([("s", "l")], 70, 10, 0, lgd_max=4096) as eng:
    eng.set_reference(ref)
    db = eng.upload(b, packed=True)
    for _ in range(3):
        eng.tabulate(db)
    eng.sync()
    # (round 6: the packed kernels run as 256 blocks of 16 wavefronts; NB / WPB in the environment for other launches)
    NB, WPB = int(os.environ.get("NB", 256)), int(os.environ.get("WPB", 16))
    nw = NB * WPB
    out = np.zeros(nw * 3, np.uint64)
    lib = engine._lib
    rc = lib.mdx_dbg_clk_read(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nw))
    t = out.reshape(nw, 3).astype(np.int64)
    t0 = t[:, 0].min()
    s, m, e = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0     # microseconds
    print("rc", rc, "waves", nw)
    for name, v in (("start", s), ("end of tile loop", m), ("end", e), ("tile loop duration", m - s), ("lists duration", e - m), ("total duration", e - s)):
        print("%-20s min %8.1f  p50 %8.1f  p90 %8.1f  p99 %8.1f  max %8.1f  mean %8.1f us" % (name, v.min(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max(), v.mean()))
    eb = e.reshape(NB, WPB).max(axis=1)
    print("block end           min %8.1f  p50 %8.1f  max %8.1f" % (eb.min(), np.percentile(eb, 50), eb.max()))
    db.free()
    d = (e - s).reshape(NB, WPB)
    bm = d.mean(axis=1)
    print("by XCD (block % 8):", " ".join("%.0f" % bm[x::8].mean() for x in range(8)))
    print("by XCD spread (std of block means within XCD):", " ".join("%.0f" % bm[x::8].std() for x in range(8)))
    print("within-block std of wave durations: mean %.1f" % d.std(axis=1).mean())
    print("block means: min %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f" % (bm.min(), np.percentile(bm, 10), np.percentile(bm, 50), np.percentile(bm, 90), bm.max()))
    order = np.argsort(bm)
    print("slowest blocks:", order[-16:], "fastest:", order[:16])
    print("block mean by block index / 64:", " ".join("%.0f" % bm[k * 64:(k + 1) * 64].mean() for k in range(8)))
    print("first 32 block means:", " ".join("%.0f" % v for v in bm[:32]))
