"""--min-basequal through the packed masked kernel (the mask folded into the resident column): kernel ms per launch at -Q 0
and -Q 20, config-3 records with 5 % of the bases below Phred 20.  MDX_LIB=<library> python tools/experiments/mq_pk.py [records]"""
import os, sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from mapdamage_amd import engine, synth
if os.environ.get('MDX_LIB'):
    engine._lib = engine.load_library(os.environ['MDX_LIB'])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
ref = synth.make_genome()
b = synth.parallel_batch(dict(read_len=100, paired=True, contigs=[0, 1], frac_softclip=0.10, frac_ins=0.04, frac_del=0.04, frac_skip=0.002,
                              frac_hardclip=0.001, with_qual=True), ref, n, 3, workers=64)
rng = np.random.default_rng(9)
low = rng.random(b.qual.shape[0]) < 0.05
b.qual = np.where(low, rng.integers(2, 20, b.qual.shape[0]), rng.integers(30, 42, b.qual.shape[0])).astype(np.uint8)
out = {}
for q in (0, 20):
    with engine.DamageEngine([("s", "l")], 70, 10, q, lgd_max=4096) as eng:
        eng.set_reference(ref)
        db = eng.upload(b, packed=True)
        eng.tabulate(db); eng.sync()
        eng.timing(True)
        for _ in range(20):
            eng.tabulate(db)
        eng.sync()
        nl, ms = eng.timing_read()
        out[q] = ms / nl
        db.free()
print("records %d: -Q 0 %.4f ms, -Q 20 %.4f ms, ratio %.3f" % (n, out[0], out[20], out[20] / out[0]))
