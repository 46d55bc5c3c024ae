import sys, pathlib, tempfile, numpy as np
sys.path.insert(0, "/root/repo")
from tests import test_rescale as T
from mapdamage_amd.engine import DamageEngine
tmp = pathlib.Path(tempfile.mkdtemp())
ref, batch, model, corr_prob, want_qual, want_mr = T.load(tmp)
with DamageEngine([("s", "l")]) as eng:
    eng.set_reference(ref); eng.set_rescale_model(model)
    q, mr, st = eng.rescale(batch)
bad = np.nonzero(q != want_qual)[0]
print("n bad", len(bad))
recs = np.searchsorted(batch.seq_off, bad, side="right") - 1
import collections
c = collections.Counter(recs.tolist())
print("records affected", len(c))
for r, n in list(c.items())[:8]:
    s0, s1 = int(batch.seq_off[r]), int(batch.seq_off[r+1])
    cg = batch.cigar[batch.cigar_off[r]:batch.cigar_off[r+1]]
    idx = bad[recs == r] - s0
    print("rec", r, "flag", hex(int(batch.flag[r])), "lseq", s1 - s0, "cigar", [(int(x) & 15, int(x) >> 4) for x in cg], "status", st[r], "bad idx", idx[:12].tolist(), "got", q[s0 + idx[:6]].tolist(), "want", want_qual[s0 + idx[:6]].tolist(), "in", batch.qual[s0 + idx[:6]].tolist())
