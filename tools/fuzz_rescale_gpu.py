"""Rounds of fuzzed records (tools/fuzz_vs_reference.py: every CIGAR shape, junk bases, missing qualities, contig edges)
and of short gapped records (tests/test_rescale.py short_records) through the rescaling kernels against the C
oracle, with models of random window lengths: qualities, MR sums, routing and the summary words.
Run on the GPU box: python tools/fuzz_rescale_gpu.py [rounds]"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.batch import batch_from_records  # noqa: E402
from mapdamage_amd.engine import DamageEngine  # noqa: E402
from mapdamage_amd.rescale import RescaleModel  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.test_rescale import corr_table, one_pass, short_records, summary_ints_from_oracle  # noqa: E402
from tests.util import assert_tables_equal, oracle_tableset  # noqa: E402
from tools.fuzz_vs_reference import fuzz_records, rescale_writable  # noqa: E402


def main():
    import torch
    torch.cuda.init()      # (before the engine: the two share the device)
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500, lower_run=3000)
    bad = 0
    for k in range(rounds):
        rng = np.random.default_rng(9000 + k)
        l5, l3 = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        corr_prob = {}
        for p in list(range(1, l5 + 1)) + list(range(-l3, 0)):
            corr_prob[("C", "T", p)] = 0.0 if rng.random() < 0.1 else float(rng.random() * 0.7)
            corr_prob[("G", "A", p)] = 0.0 if rng.random() < 0.1 else float(rng.random() * 0.7)
        model = RescaleModel(corr_prob, l5, l3)
        if k % 2 == 0:
            recs = [r for r in fuzz_records(ref, 6000, 7000 + k, with_qual=True) if rescale_writable(r["cigar"])]
        else:
            recs = short_records(ref, 7000 + k, n=8000)
        b = batch_from_records(recs, with_qual=True)
        b.mtid = np.where(rng.random(b.n) < 0.9, b.tid, (b.tid + 1) % 2).astype(np.int32)
        b.mpos = (b.pos + rng.integers(-300, 300, size=b.n)).astype(np.int32)
        want_q, want_mr, want_st, want_counts, _ = oracle.rescale_with_subs(ref, b, corr_table(corr_prob, model), l5, l3)
        with DamageEngine([("s", "l")]) as eng:
            eng.set_reference(ref)
            eng.set_rescale_model(model)
            got_q, got_mr, got_st = eng.rescale(b)
            words = eng.rescale_summary()
        ok = (np.array_equal(got_q, want_q) and np.array_equal(got_st, want_st) and np.array_equal(np.isnan(got_mr), np.isnan(want_mr))
              and np.array_equal(got_mr[~np.isnan(got_mr)], want_mr[~np.isnan(want_mr)])
              and np.array_equal(words[:756], summary_ints_from_oracle(want_counts)))
        # the same batch through the fused launch (mdx_tabulate_rescale_device) at a random --length: tables too
        length = int(rng.integers(8, 125))
        libs = [("s", "l")]
        try:
            want_tables = oracle_tableset(ref, b, libs, length, 10, 0)
        except Exception as e:      # (fuzzed batches hold records the tabulation rejects: then only the rescaling is compared)
            want_tables = None
        okf = True
        for packed in (False, True):       # (both forms of the SEQ column: the fused ASCII kernel, the packed fused kernel)
            with DamageEngine(libs, length, 10, 0) as eng:
                eng.set_reference(ref)
                eng.set_rescale_model(model)
                try:
                    fq, fmr, fst = one_pass(eng, b, packed)
                    fwords = eng.rescale_summary()
                    ftables = eng.finish() if want_tables is not None else None
                    okf1 = (np.array_equal(fq, want_q) and np.array_equal(fst, want_st) and np.array_equal(np.isnan(fmr), np.isnan(want_mr))
                           and np.array_equal(fmr[~np.isnan(fmr)], want_mr[~np.isnan(want_mr)]) and np.array_equal(fwords, words))
                    if ftables is not None:
                        try:
                            assert_tables_equal(ftables, want_tables)
                        except AssertionError:
                            okf1 = False
                except Exception as e:
                    okf1 = want_tables is None
                    print("   fused pass (%s) raised:" % ("4-bit" if packed else "ASCII"), type(e).__name__, str(e)[:100])
            if not okf1:
                print("   fused pass (%s): MISMATCH" % ("4-bit" if packed else "ASCII"))
            okf = okf and okf1
        ok = ok and okf
        print("round %d (%s, model %d+%d, %d records): %s" % (k, "fuzzed CIGARs" if k % 2 == 0 else "short records", l5, l3, b.n,
                                                              "equal" if ok else "MISMATCH"), flush=True)
        bad += not ok
    print("%d rounds, %d mismatches" % (rounds, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
