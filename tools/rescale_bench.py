"""Kernel-only throughput of the quality-rescaling pass (BASELINE configs[4] / survey config 5: config-3
records with Phred 2..41 qualities, fixed correction-probability CSV), checked against the C oracle.
Run on the GPU box: python tools/rescale_bench.py [reads]"""
import json
import pathlib
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.engine import DamageEngine  # noqa: E402
from mapdamage_amd.rescale import RescaleModel, get_corr_prob  # noqa: E402
from oracle import oracle  # noqa: E402
from tools.make_golden import rescale_csv  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    plain = len(sys.argv) > 2 and sys.argv[2] == "plain"    # no indels / skips: fast path only
    ref = synth.make_genome()
    # config 3 without hard clips: an H next to an S leaves the reference with a quality string of the wrong
    # length (rescale.py:266-271), which it cannot write
    g = 0.0 if plain else 1.0
    b = synth.make_reads(ref, n, 5, read_len=100, paired=True, frac_softclip=0.10, frac_ins=0.04 * g, frac_del=0.04 * g,
                         frac_skip=0.002 * g, frac_hardclip=0.0, contigs=[0, 1], with_qual=True)
    rng = np.random.default_rng(5)
    b.mtid = b.tid.copy()
    b.mpos = (b.pos + rng.integers(-300, 300, size=b.n)).astype(np.int32)
    with tempfile.TemporaryDirectory() as tmp:
        path = pathlib.Path(tmp) / "Stats_out_MCMC_correct_prob.csv"
        path.write_text(rescale_csv())
        model = RescaleModel.from_csv(path, 12, 12)
        cp = get_corr_prob(path, 12, 12)
    corr = np.zeros((2, model.npos))
    for (r, _s, p), v in cp.items():
        corr[0 if r == "C" else 1, p if p > 0 else model.len5p - p] = v
    with DamageEngine([("s", "l")]) as eng:
        eng.set_reference(ref)
        eng.set_rescale_model(model)
        eng.rescale(b)                      # warm-up
        eng.timing(True)
        reps = 5
        for _ in range(reps):
            q, mr, st = eng.rescale(b)
        n_launch, ms = eng.rescale_timing_read()
    t0 = time.perf_counter()
    wq, wmr, wst = oracle.rescale(ref, b, corr, model.len5p, model.len3p)
    cpu = time.perf_counter() - t0
    ok = (np.array_equal(q, wq) and np.array_equal(st, wst) and np.array_equal(np.isnan(mr), np.isnan(wmr)) and np.array_equal(mr[~np.isnan(mr)], wmr[~np.isnan(wmr)]))
    per = ms / n_launch
    # survey 8d: config 5 adds qlen (quality in) + qlen (quality out) to the 240-ish bytes of a record
    algo = float(4 * b.seq.shape[0] + 4 * b.cigar.shape[0] + (2 * 10 + 16) * b.n)
    print(json.dumps({"workload": "survey config 5: %d config-3 records with qualities, rescale kernel only" % n,
                      "kernel_ms": per, "reads_per_s": n / (per * 1e-3), "rescaled_records": int((st >= 2).sum() - (st == 4).sum()),
                      "algorithmic_GBps": algo / (per * 1e-3) / 1e9, "frac_of_8TBps": algo / (per * 1e-3) / 8e12,
                      "parity": "bit-exact vs oracle" if ok else "MISMATCH",
                      "oracle_1thread_reads_per_s": n / cpu}))


if __name__ == "__main__":
    main()
