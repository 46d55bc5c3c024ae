"""Golden fixture for rescaling records whose CIGAR carries hard clips: the reference's own
_rescale_qual_core (tools/ref_harness.py, build container only) over 50M5H / 5H50M / 5H50M5H / clip + indel shapes.
Writes tests/golden/genome_rescale_hardclip.npz.  usage: python tools/make_golden_hardclip.py"""
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.batch import batch_from_records  # noqa: E402


def hardclip_records(ref, n=600, seed=5):
    """Records whose CIGAR ends in a hard clip (bwa-mem supplementary alignments; rescaling applies no flag filter):
    50M5H, 5H50M, 5H50M5H, also around a soft clip on the other side and with an indel."""
    rng = np.random.default_rng(seed)
    bases, offs = ref.concat()
    upper = bases & np.uint8(0xDF)
    shapes = [[(0, 50), (5, 5)], [(5, 5), (0, 50)], [(5, 5), (0, 50), (5, 5)], [(4, 4), (0, 40), (5, 7)],
              [(5, 3), (0, 30), (1, 2), (0, 20), (5, 9)], [(5, 6), (0, 25), (2, 3), (0, 25), (4, 5)]]
    recs = []
    for i in range(n):
        cig = shapes[i % len(shapes)]
        span = sum(ln for op, ln in cig if op in (0, 2))
        pos = int(rng.integers(20, ref.lengths[0] - span - 20))
        seq, r = [], int(offs[0]) + pos
        for op, ln in cig:
            if op == 0:
                seq.append(upper[r:r + ln].copy()); r += ln
            elif op == 2:
                r += ln
            elif op in (1, 4):
                seq.append(rng.choice(np.frombuffer(b"ACGT", np.uint8), ln))
        seq = np.concatenate(seq)
        # damage-like substitutions so that something is rescaled
        seq = np.where((seq == ord("C")) & (rng.random(seq.shape[0]) < 0.3), ord("T"), seq)
        seq = np.where((seq == ord("G")) & (rng.random(seq.shape[0]) < 0.3), ord("A"), seq).astype(np.uint8)
        recs.append(dict(flag=int(rng.choice([0, 16, 0x800, 0x810])), tid=0, pos=pos, cigar=cig, seq=seq.tobytes().decode(),
                         qual=rng.integers(2, 42, seq.shape[0]).astype(np.uint8), lib=0, tlen=0))
    return recs


def main():
    from tools import make_golden, ref_harness
    ref = synth.small_genome()
    b = batch_from_records(hardclip_records(ref), with_qual=True)
    b.mtid = b.tid.copy()
    b.mpos = b.pos.copy()
    csv_text = make_golden.rescale_csv()
    quals, mrs, _log = ref_harness.run_reference_rescale(ref, b, csv_text, 12, 10)
    qflat = b.qual.copy()
    for i, q in enumerate(quals):
        s0, s1 = int(b.seq_off[i]), int(b.seq_off[i + 1])
        assert q is not None and len(q) == s1 - s0
        qflat[s0:s1] = np.asarray(q, dtype=np.uint8)
    out = ROOT / "tests" / "golden" / "genome_rescale_hardclip.npz"
    np.savez_compressed(out, ref_bases=np.frombuffer(b"".join(ref.seqs), dtype=np.uint8),
                        ref_lengths=np.asarray(ref.lengths, dtype=np.int64),
                        names=np.frombuffer(json.dumps(ref.names).encode(), dtype=np.uint8),
                        flag=b.flag, tid=b.tid, pos=b.pos, tlen=b.tlen, cigar_off=b.cigar_off, cigar=b.cigar,
                        seq_off=b.seq_off, seq=b.seq, qual=b.qual, mtid=b.mtid, mpos=b.mpos,
                        csv=np.frombuffer(csv_text.encode(), dtype=np.uint8), len5p=12, len3p=10,
                        qual_out=qflat, mr=np.asarray([np.nan if m is None else m for m in mrs], dtype=np.float64))
    print("genome_rescale_hardclip: reads=%d rescaled=%d changed qualities=%d" % (b.n, sum(m is not None for m in mrs), int((qflat != b.qual).sum())))


if __name__ == "__main__":
    main()
