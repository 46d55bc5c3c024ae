"""Long fuzz of the HIP path against the C oracle: random window geometry, quality threshold, number of libraries,
records from tools/fuzz_vs_reference.py plus ordinary ones.  Run on the GPU box: python tools/fuzz_gpu.py [rounds]"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.batch import batch_from_records, concat_batches  # noqa: E402
from mapdamage_amd.engine import DamageEngine  # noqa: E402
from tools.fuzz_vs_reference import fuzz_records  # noqa: E402


def main():
    from oracle import oracle
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500, lower_run=3000)
    rng = np.random.default_rng(77)
    bad = 0
    for k in range(rounds):
        L = int(rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 25, 50, 70, 71, 100, 150, 200, 238, 240, 249, 400]))
        A = int(rng.choice([0, 1, 3, 7, 8, 9, 10, 16, 30, 100]))
        Q = int(rng.choice([0, 0, 10, 20, 35]))
        nlib = int(rng.choice([1, 2, 3, 5]))
        wq = Q > 0 or rng.random() < 0.3
        recs = fuzz_records(ref, 3000, 20000 + k, with_qual=wq)
        for r in recs:
            r["lib"] = int(rng.integers(0, nlib))
        fuzz = batch_from_records(recs, with_qual=True if wq else None)
        plain = synth.make_reads(ref, 20000, 300 + k, len_range=(20, 180), nlib=nlib, frac_softclip=0.2, frac_ins=0.05,
                                 frac_del=0.05, frac_skip=0.01, with_qual=wq, paired=bool(k % 2), clip_max=30)
        batch = concat_batches([plain, fuzz])
        libs = [("s", "l%d" % i) for i in range(nlib)]
        want = oracle.tabulate(ref, batch, nlib, L, A, Q, 65536)
        with DamageEngine(libs, L, A, Q) as eng:
            eng.set_reference(ref)
            eng.tabulate(batch)
            got = eng.finish()
            mode = eng.table_mode
        ok = (np.array_equal(got.mis, want["mis"]) and np.array_equal(got.comp, want["comp"])
              and np.array_equal(got.lgd, want["lgd"]) and got.n_kept == want["n_kept"])
        print("round %d L=%d A=%d Q=%d nlib=%d qual=%s mode=%s : %s" % (k, L, A, Q, nlib, wq, mode, "equal" if ok else "DIFFERENT"), flush=True)
        bad += not ok
    print("rounds with differences:", bad)
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
