import sys, pathlib, tempfile
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from mapdamage_amd import synth
from mapdamage_amd.batch import batch_from_records
from mapdamage_amd.engine import DamageEngine
from oracle import oracle
from tools.fuzz_vs_reference import fuzz_records, rescale_writable
from tests.test_rescale import load, corr_table, summary_ints_from_oracle
k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
with tempfile.TemporaryDirectory() as tmp:
    _, _, model, corr_prob, _, _ = load(pathlib.Path(tmp))
ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500, lower_run=3000)
recs = [r for r in fuzz_records(ref, 5000, 8100 + k, with_qual=True) if rescale_writable(r["cigar"])]
sub = None
def run(recs):
    b = batch_from_records(recs, with_qual=True)
    rng = np.random.default_rng(30 + k)
    b.mtid = np.where(rng.random(b.n) < 0.9, b.tid, (b.tid + 1) % 2).astype(np.int32)
    b.mpos = (b.pos + rng.integers(-300, 300, size=b.n)).astype(np.int32)
    want_q, want_mr, want_st, want_counts, want_pvals = oracle.rescale_with_subs(ref, b, corr_table(corr_prob, model), model.len5p, model.len3p)
    with DamageEngine([("s", "l")]) as eng:
        eng.set_reference(ref); eng.set_rescale_model(model)
        got_q, got_mr, got_st = eng.rescale(b)
        words = eng.rescale_summary()
    w = summary_ints_from_oracle(want_counts)
    d = np.nonzero(words[:756] != w)[0]
    return b, d, words, w, (got_q != want_q).sum(), (got_st != want_st).sum(), got_mr, want_mr
b, d, words, w, nq, nst, gmr, wmr = run(recs)
print("summary diffs", len(d), [(int(i), int(words[i]), int(w[i])) for i in d[:12]], "q diffs", nq, "st diffs", nst,
      "mr diffs", int((~np.isnan(wmr) & (gmr != wmr)).sum()))
# bisect to one record
lo, hi = 0, len(recs)
cur = recs
while len(cur) > 1:
    half = len(cur) // 2
    a1, a2 = cur[:half], cur[half:]
    r1 = run(a1)
    if len(r1[1]) or r1[4]:
        cur = a1
    else:
        cur = a2
r = run(cur)
print("single record:", cur[0]["cigar"], cur[0]["flag"], cur[0]["pos"], len(cur[0]["seq"]), "diffs", [(int(i), int(r[2][i]), int(r[3][i])) for i in r[1][:8]], "q", r[4])
print(cur[0]["seq"])
