"""Fuzz of the rescaling pass: the reference's own _rescale_qual_core (tools/ref_harness.py, build container only)
against the C oracle on the random CIGARs of tools/fuzz_vs_reference.py (hard clips only where the reference can write the record: not outside a soft clip): new qualities and MR tags.
usage: python tools/fuzz_rescale_vs_reference.py [rounds]"""
import pathlib
import sys
import tempfile

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.batch import batch_from_records  # noqa: E402
from mapdamage_amd.rescale import RescaleModel, get_corr_prob  # noqa: E402
from tools.fuzz_vs_reference import fuzz_records, rescale_writable  # noqa: E402


def main():
    from oracle import oracle
    from tools import make_golden, ref_harness
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500, lower_run=3000)
    csv_text = make_golden.rescale_csv()
    with tempfile.TemporaryDirectory() as tmp:
        path = pathlib.Path(tmp) / "Stats_out_MCMC_correct_prob.csv"
        path.write_text(csv_text)
        model = RescaleModel.from_csv(path, 12, 10)
        cp = get_corr_prob(path, 12, 10)
    corr = np.zeros((2, model.npos))
    for (r, _s, p), v in cp.items():
        corr[0 if r == "C" else 1, p if p > 0 else model.len5p - p] = v
    for k in range(rounds):
        recs = [r for r in fuzz_records(ref, 1200, 9100 + k, with_qual=True) if rescale_writable(r["cigar"])]
        b = batch_from_records(recs, with_qual=True)
        rng = np.random.default_rng(k)
        b.mtid = np.where(rng.random(b.n) < 0.9, b.tid, (b.tid + 1) % 2).astype(np.int32)
        b.mpos = (b.pos + rng.integers(-300, 300, size=b.n)).astype(np.int32)
        quals, mrs, _log = ref_harness.run_reference_rescale(ref, b, csv_text, 12, 10)
        got_q, got_mr, _st = oracle.rescale(ref, b, corr, model.len5p, model.len3p)
        bad = 0
        for i in range(b.n):
            s0, s1 = int(b.seq_off[i]), int(b.seq_off[i + 1])
            if quals[i] is not None and list(got_q[s0:s1]) != list(quals[i]):
                bad += 1
            want = mrs[i]
            have = None if np.isnan(got_mr[i]) else float("%.5f" % got_mr[i])
            if (want is None) != (have is None) or (want is not None and want != have):
                bad += 1
        print("round %d records=%d : %s" % (k, b.n, "equal" if bad == 0 else "%d DIFFERENT" % bad))
        if bad:
            raise SystemExit(1)


if __name__ == "__main__":
    main()
