"""tools/experiments/phase_clk.py for the fused tabulate + rescale launch (config 5).
    tools/mkvariant.sh phclk -DMDX_WAVE_CLK -DMDX_PHASE_CLK
    gpurun -- 'MDX_LIB=tools/bin/libmdx_phclk.so python tools/experiments/phase_clk5.py 10000000'"""
import ctypes, os, sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
os.environ["MDX_DBG_CLK"] = "1"
import torch
torch.cuda.init()
from mapdamage_amd import engine, synth
import bench
if os.environ.get('MDX_LIB'):
    engine._lib = engine.load_library(os.environ['MDX_LIB'])
NAMES = ["other (tile hand-out, rounds)", "phase 1 of a tile", "complete runs of the tile", "partial runs of the tile", "drain (events)",
         "planes -> LDS", "general pass", "lists: staging", "lists: staging (insertions)", "lists: staging (deletions)",
         "final drain + fold", "qualities of the listed transitions (rsq_flush)", "general: walks (phase 2b)", "general: compositions behind deletions",
         "lists: the runs themselves", "the tile's copy of the quality column"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ref = synth.make_genome()
model, corr = bench.rescale_fixture()
b = bench.add_mates(synth.parallel_batch(bench.CONFIG5, ref, n, 3, workers=64), 7)
dev = torch.device("cuda", 0)
lib = engine._lib
with engine.DamageEngine([("s", "l")], 70, 10, 0, lgd_max=4096) as eng:
    eng.set_reference(ref)
    eng.set_rescale_model(model)
    db = eng.upload(b, packed=True)
    rs = bench.RescaleBuffers(torch, dev, b)
    rs.run(eng, db)
    eng.sync()
    out = np.zeros(16, np.uint64)
    lib.mdx_dbg_phase_read(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(1))
    K = 5
    import time
    t0 = time.perf_counter()
    for _ in range(K):
        rs.run(eng, db)
    eng.sync()
    dt = (time.perf_counter() - t0) / K
    lib.mdx_dbg_phase_read(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(1))
    t = out.astype(np.float64) / K
    print("== config 5: %d records, %.3f ms per call (instrumented, wall), ticks per record %.1f; fused launches %s" % (n, dt * 1e3, t.sum() / n, eng.fused_launches()))
    for name, v in zip(NAMES, t):
        print("   %-50s %6.2f %%   %7.2f ticks per record" % (name, 100.0 * v / t.sum(), v / n))
    db.free()
