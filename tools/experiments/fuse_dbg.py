import os, sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch
torch.cuda.init()
from mapdamage_amd import synth
from mapdamage_amd.engine import DamageEngine, BadReadError
from mapdamage_amd.rescale import RescaleModel
from tests.test_rescale import one_pass, corr_table
length, l5, l3, lens = 70, 40, 3, (5, 90)
rng = np.random.default_rng(100 + length)
corr_prob = {}
for p in list(range(1, l5 + 1)) + list(range(-l3, 0)):
    corr_prob[("C", "T", p)] = float(rng.random() * 0.7)
    corr_prob[("G", "A", p)] = float(rng.random() * 0.7)
model = RescaleModel(corr_prob, l5, l3)
ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500, lower_run=3000)
b = synth.make_reads(ref, 70_000, 40 + length, len_range=lens, paired=True, frac_softclip=0.2, frac_ins=0.05,
                     frac_del=0.05, frac_skip=0.01, with_qual=True, frac_filtered=0.03)
b.mtid = np.where(rng.random(b.n) < 0.9, b.tid, (b.tid + 1) % 2).astype(np.int32)
b.mpos = (b.pos + rng.integers(-300, 300, size=b.n)).astype(np.int32)
b.flag = np.where(rng.random(b.n) < 0.4, b.flag & 0xF14, b.flag).astype(np.uint16)
b.flag = np.where(rng.random(b.n) < 0.05, b.flag | 0x400, b.flag).astype(np.uint16)
for fuse in ("0", "1"):
    os.environ["MDX_NO_FUSE"] = fuse
    with DamageEngine([("s", "l")], length, 10, 0) as eng:
        eng.set_reference(ref); eng.set_rescale_model(model)
        try:
            one_pass(eng, b); print("NO_FUSE", fuse, "ok")
        except BadReadError as e:
            i = e.args[0] if isinstance(e.args[0], int) else 1253
            print("NO_FUSE", fuse, "bad read", e)
            for i in (1253,):
                co = b.cigar_off[i], b.cigar_off[i+1]
                print(i, "flag", hex(b.flag[i]), "tid", b.tid[i], "pos", b.pos[i], "lseq", b.seq_off[i+1]-b.seq_off[i],
                      "cigar", [(int(c) >> 4, int(c) & 15) for c in b.cigar[co[0]:co[1]]])
