"""Fuzz: random CIGARs (M I D N P = X inside, S / H at the ends, 1-9 operations), flags, insert sizes, qualities and
libraries, tabulated by the reference itself (build container only: imports /root/reference through
tools/ref_harness.py) and by the C oracle; tables and the three text files must be equal.
usage: python tools/fuzz_vs_reference.py [rounds] [--save NAME]  (--save writes the last round as a golden fixture)"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.batch import batch_from_records  # noqa: E402
from mapdamage_amd.tables import TableSet, merge_library_ids  # noqa: E402


def fuzz_records(ref, n, seed, with_qual):
    rng = np.random.default_rng(seed)
    bases, offs = ref.concat()
    upper = bases & np.uint8(0xDF)
    lens = list(ref.lengths)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    recs = []
    while len(recs) < n:
        tid = int(rng.integers(0, 2))
        nops = int(rng.integers(1, 8))
        ops = []
        for k in range(nops):
            if k % 2 == 0:
                ops.append((int(rng.choice([0, 0, 0, 7, 8])), int(rng.integers(1, 60))))
            else:
                op = int(rng.choice([1, 2, 3, 6, 1, 2]))
                ops.append((op, int(rng.integers(1, 40 if op != 3 else 200))))
        if ops[-1][0] not in (0, 7, 8):
            ops.append((0, int(rng.integers(1, 60))))
        if rng.random() < 0.1:      # adjacent insertion + deletion
            ops.insert(1, (2, int(rng.integers(1, 5))))
            ops.insert(1, (1, int(rng.integers(1, 5))))
        span = sum(ln for op, ln in ops if op in (0, 2, 3, 7, 8))
        edge = rng.random()
        if edge < 0.03:
            pos = int(rng.integers(0, 8))
        elif edge < 0.06:
            pos = lens[tid] - span - int(rng.integers(0, 8))
        else:
            pos = int(rng.integers(0, lens[tid] - span))
        seq, r = [], offs[tid] + pos
        for op, ln in ops:
            if op in (0, 7, 8):
                seq.append(upper[r:r + ln].copy()); r += ln
            elif op == 1:
                seq.append(rng.choice(acgt, ln))
            elif op in (2, 3):
                r += ln
        sl, sr = (int(rng.integers(1, 12)) if rng.random() < 0.25 else 0 for _ in range(2))
        seq = np.concatenate([rng.choice(acgt, sl)] + seq + [rng.choice(acgt, sr)])
        seq = np.where(rng.random(seq.shape[0]) < 0.06, rng.choice(np.frombuffer(b"ACGTNacgtRY", np.uint8), seq.shape[0]), seq)
        cig = ([(5, 3)] if rng.random() < 0.08 else []) + ([(4, sl)] if sl else []) + ops + \
              ([(4, sr)] if sr else []) + ([(5, 2)] if rng.random() < 0.08 else [])
        flag = int(rng.choice([0, 16]))
        tlen = 0
        if rng.random() < 0.5:      # paired
            flag |= 0x1 | int(rng.choice([0x40, 0x80])) | (0x2 if rng.random() < 0.8 else 0) | (0x20 if rng.random() < 0.5 else 0)
            tlen = int(rng.integers(-700, 700))
        if rng.random() < 0.05:
            flag |= int(rng.choice([0x4, 0x100, 0x200, 0x400, 0x800]))
        qual = None
        if with_qual and rng.random() > 0.04:
            qual = np.where(rng.random(seq.shape[0]) < 0.2, rng.integers(0, 20, seq.shape[0]),
                            rng.integers(20, 42, seq.shape[0])).astype(np.uint8)
        recs.append(dict(flag=flag, tid=tid, pos=pos, cigar=cig, seq=seq.tobytes().decode(), qual=qual,
                         lib=int(rng.integers(0, 3)), tlen=tlen))
    return recs


def rescale_writable(cigar):
    """rescale.py:266-271 re-attaches the clipped qualities only when the first / last operation is S: a hard clip
    outside a soft clip (H S ... / ... S H) leaves a quality string shorter than SEQ, which pysam refuses to
    write.  Hard clips without a soft clip next to them (5H50M, 50M5H, 5H50M5H) are rescaled like any record."""
    ops = [op for op, _ in cigar]
    return not ((len(ops) > 1 and ops[0] == 5 and ops[1] == 4) or (len(ops) > 1 and ops[-1] == 5 and ops[-2] == 4))


def main():
    from oracle import oracle
    from tools import make_golden, ref_harness
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
    save = sys.argv[sys.argv.index("--save") + 1] if "--save" in sys.argv else None
    ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500, lower_run=3000)
    libs = [("Zed", "libB"), ("Alpha", "libA"), ("Mid", "libC")]
    rng = np.random.default_rng(2024)
    for k in range(rounds):
        L, A = [(70, 10), (8, 3), (25, 5), (150, 30), (1, 0), (100, 12)][k % 6]
        Q = int(rng.choice([0, 15, 20, 30])) if k % 2 else 0
        batch = batch_from_records(fuzz_records(ref, 1500, 5000 + k, with_qual=Q > 0), with_qual=True if Q > 0 else None)
        res = ref_harness.run_reference(ref, batch, libs, L, A, Q)
        slibs, mis, comp, lgd = ref_harness.dense_tables(res, libs, L, A)
        got = oracle.tabulate(ref, batch, len(libs), L, A, Q, 65536)
        ts = TableSet(list(libs), L, A, got["mis"], got["comp"], got["lgd"], got["lgd_over"], got["n_kept"])
        order = [libs.index(lib) for lib in slibs]
        ok = (np.array_equal(ts.mis[order], mis) and np.array_equal(ts.comp[order], comp) and ts.n_kept == res["n_kept"]
              and ts.misincorporation_text() == res["texts"]["misincorporation.txt"]
              and ts.dnacomp_text() == res["texts"]["dnacomp.txt"]
              and ts.lgdistribution_text() == res["texts"]["lgdistribution.txt"])
        print("round %d L=%d A=%d Q=%d records=%d kept=%d : %s" % (k, L, A, Q, batch.n, res["n_kept"], "equal" if ok else "DIFFERENT"))
        if not ok:
            raise SystemExit(1)
        if save and k == rounds - 1:
            make_golden.save_case(save, ref, batch, libs, L, A, Q)


if __name__ == "__main__":
    main()
