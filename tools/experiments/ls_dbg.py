import ctypes, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from mapdamage_amd import synth
from mapdamage_amd.engine import DamageEngine, MdxBatch, MdxError
ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500, lower_run=3000)
batch = synth.make_reads(ref, 20_000, 23, len_range=(30, 120), nlib=2, paired=True, frac_softclip=0.1, frac_ins=0.04, frac_del=0.04)
libs2, libs3 = [("s", "a"), ("s", "b")], [("s", "a"), ("s", "b"), ("s", "c")]
def say(x): print(x, flush=True)
eng2 = DamageEngine(libs2, 70, 10, 0); eng3 = DamageEngine(libs3, 70, 10, 0)
eng2.set_reference(ref); eng3.set_reference(ref)
db = eng2.upload(batch, packed=True); say("uploaded")
eng2.tabulate(db); eng2.finish(); say("tabulated")
view = MdxBatch(); ctypes.memmove(ctypes.byref(view), ctypes.byref(db.dev), ctypes.sizeof(MdxBatch))
try: eng3.tabulate_view(view)
except MdxError as e: say("eng3 refused: %s" % e)
if "half" in sys.argv:
    view.n_reads = batch.n // 2; view.n_cigar = int(batch.cigar_off[batch.n // 2]); view.n_bases = int(batch.seq_off[batch.n // 2])
    try: eng2.tabulate_view(view)
    except MdxError as e: say("eng2 refused: %s" % e)
    view.libsort = None
    eng2.reset(); eng2.tabulate_view(view); eng2.finish(); say("half view counted")
db.free(); say("freed")
eng3.close(); say("eng3 closed")
eng2.close(); say("eng2 closed")
