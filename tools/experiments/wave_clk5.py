"""wave_clk.py for the fused tabulate + rescale kernel (config 5): per-wavefront clocks of its 256 x 16 wavefronts.
Needs the instrumented build (tools/mkvariant.sh clk -DMDX_WAVE_CLK; MDX_LIB=tools/bin/libmdx_clk.so)."""
import ctypes, os, sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
os.environ["MDX_DBG_CLK"] = "1"
import torch
torch.cuda.init()
from mapdamage_amd import engine, synth
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
ref = synth.make_genome()
model, corr = bench.rescale_fixture()
b = bench.add_mates(synth.parallel_batch(bench.CONFIG5, ref, n, 3, workers=64), 7)
dev = torch.device("cuda", 0)
with engine.DamageEngine([("s", "l")], 70, 10, 0, lgd_max=4096) as eng:
    eng.set_reference(ref)
    eng.set_rescale_model(model)
    db = eng.upload(b)
    rs = bench.RescaleBuffers(torch, dev, b)
    for _ in range(3):
        rs.run(eng, db)
    eng.sync()
    print("fused launches", eng.fused_launches())
    nw = 256 * 16
    out = np.zeros(nw * 3, np.uint64)
    rc = engine._lib.mdx_dbg_clk_read(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nw))
    t = out.reshape(nw, 3).astype(np.int64)
    t0 = t[:, 0].min()
    s, m, e = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0
    for name, v in (("end of tile loop", m), ("end", e), ("lists duration", e - m)):
        print("%-20s min %8.1f  p10 %8.1f  p50 %8.1f  p90 %8.1f  p99 %8.1f  max %8.1f  mean %8.1f us" % (name, v.min(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max(), v.mean()))
    d = e.reshape(256, 16)
    print("within-block std of wave ends: mean %.1f; block-mean ends: min %.0f p50 %.0f max %.0f" % (d.std(axis=1).mean(), d.mean(axis=1).min(), np.percentile(d.mean(axis=1), 50), d.mean(axis=1).max()))
    print("by wave slot within block (mean end):", " ".join("%.0f" % v for v in d.mean(axis=0)))
    db.free()
