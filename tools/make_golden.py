"""Generate tests/golden/*.npz by running the reference's own functions (build container only).

    python tools/make_golden.py            # (re)writes every fixture
    python tools/make_golden.py --time     # also times the reference's Python path

Each fixture holds the *inputs* (reference contigs + SoA batch columns + parameters) and the
*expected outputs* produced by /root/reference/mapdamage (dense tables in the canonical layout
of mapdamage_amd/layout.py, the sparse length histogram, and the three text files byte for
byte).  Fixtures are data; no reference source is stored.
"""

import argparse
import json
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.batch import Reference, batch_from_records  # noqa: E402
from tools import ref_harness  # noqa: E402

GOLDEN = ROOT / "tests" / "golden"

LIBS2 = [("Zed", "libB"), ("Alpha", "libA")]   # sorted order differs from header order
LIBS1 = [("Sample1", "Lib1")]
MERGED = [("*", "*")]


def save_case(name, ref, batch, libraries, length, around, minqual, per_read=False, note=""):
    t0 = time.perf_counter()
    res = ref_harness.run_reference(ref, batch, libraries, length, around, minqual,
                                    per_read=per_read)
    dt = time.perf_counter() - t0
    libs, mis, comp, lgd = ref_harness.dense_tables(res, libraries, length, around)
    meta = dict(name=name, length=length, around=around, minqual=minqual,
                libraries=[list(x) for x in libraries], sorted_libraries=[list(x) for x in libs],
                contig_names=ref.names, n_reads=batch.n, n_kept=res["n_kept"], note=note,
                generator="tools/make_golden.py", reference="ginolhac/mapDamage 2.3.0a0 @2024_10_08")
    arrays = dict(
        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
        ref_bases=np.frombuffer(b"".join(ref.seqs), dtype=np.uint8),
        ref_lengths=np.asarray(ref.lengths, dtype=np.int64),
        flag=batch.flag, lib=batch.lib, tid=batch.tid, pos=batch.pos, tlen=batch.tlen,
        cigar_off=batch.cigar_off, cigar=batch.cigar, seq_off=batch.seq_off, seq=batch.seq,
        mis=mis, comp=comp, lgd=lgd,
        txt_mis=np.frombuffer(res["texts"]["misincorporation.txt"].encode(), dtype=np.uint8),
        txt_comp=np.frombuffer(res["texts"]["dnacomp.txt"].encode(), dtype=np.uint8),
        txt_lgd=np.frombuffer(res["texts"]["lgdistribution.txt"].encode(), dtype=np.uint8),
    )
    if batch.qual is not None:
        arrays["qual"] = batch.qual
    if per_read:
        arrays["per_read"] = np.frombuffer(json.dumps(res["per_read"]).encode(), dtype=np.uint8)
    GOLDEN.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(GOLDEN / (name + ".npz"), **arrays)
    print("%-28s reads=%6d kept=%6d  mis_sum=%d  %.2fs" % (name, batch.n, res["n_kept"],
                                                           int(mis.sum()), dt))
    return res


def appendix_d_case():
    """The ten hand vectors of SURVEY.md Appendix D on the 60-base genome ``c``."""
    c = b"ACGTACGTTGCAAGGCTTAACCGGTTACGATCGATCGGGATATCCGCGATATAGCTAGCT"
    ref = Reference(["c"], [c])
    M, I, D, N, S, H = 0, 1, 2, 3, 4, 5
    R = 0x10
    recs = [
        dict(flag=0, tid=0, pos=1, cigar=[(M, 10)], seq="TGTACGTTGC"),
        dict(flag=R, tid=0, pos=1, cigar=[(M, 10)], seq="CGTACGTTGC"),
        dict(flag=0, tid=0, pos=4, cigar=[(S, 2), (M, 8), (S, 3)], seq="GGACGTTGCATTT"),
        dict(flag=0, tid=0, pos=0, cigar=[(M, 4), (N, 5), (M, 4)], seq="ACGTGCAA"),
        dict(flag=0, tid=0, pos=0, cigar=[(M, 3), (I, 2), (M, 3), (D, 1), (M, 2)], seq="ACGGGTACTT"),
        dict(flag=0, tid=0, pos=0, cigar=[(M, 6)], seq="ACGTAC"),
        dict(flag=0, tid=0, pos=54, cigar=[(M, 6)], seq="CTAGCT"),
        dict(flag=0, tid=0, pos=0, cigar=[(M, 3), (I, 1), (M, 3)], seq="ACGTTAC",
             qual=[40, 40, 2, 40, 2, 40, 40]),
        dict(flag=0, tid=0, pos=0, cigar=[(M, 6)], seq="ANGTRC"),
        dict(flag=0, tid=0, pos=0, cigar=[(H, 2), (M, 6)], seq="ACGTAC"),
        dict(flag=0, tid=0, pos=0, cigar=[(I, 2), (M, 6)], seq="TTACGTAC"),
        dict(flag=R, tid=0, pos=10, cigar=[(S, 1), (M, 5), (I, 1), (M, 4), (D, 2), (M, 3), (S, 2)],
             seq="GCAAGGTCTTAACGTT"),
    ]
    for r in recs:
        r.setdefault("qual", None)
        r.setdefault("lib", 0)
        r.setdefault("tlen", 0)
    return ref, batch_from_records(recs, with_qual=True)


def rescale_csv():
    """A Stats_out_MCMC_correct_prob.csv in the format of r/stats/main.r:225 (positions 1..12, -12..-1)."""
    rows = ['"","Position","C.T","G.A"']
    for i, p in enumerate(list(range(1, 13)) + list(range(-12, 0))):
        ct = 0.5 * 0.6 ** (abs(p) - 1) if p > 0 else 0.0123
        ga = 0.47 * 0.55 ** (abs(p) - 1) if p < 0 else 0.0217
        rows.append('"%d",%d,%r,%r' % (i + 1, p, ct, ga))
    return "\n".join(rows) + "\n"


def rescale_batch(ref, n=1500, seed=5):
    """Mixed single-end / paired records with qualities, mates, clips, indels, skips, filtered flags."""
    b = synth.make_reads(ref, n, seed, len_range=(20, 120), paired=True, frac_softclip=0.15, frac_ins=0.1,
                         frac_del=0.1, frac_skip=0.02, with_qual=True, frac_filtered=0.05)
    rng = np.random.default_rng(seed + 100)
    b.mtid = np.where(rng.random(b.n) < 0.9, b.tid, (b.tid + 1) % 2).astype(np.int32)
    b.mpos = (b.pos + rng.integers(-200, 200, size=b.n)).astype(np.int32)
    single = rng.random(b.n) < 0.5
    b.flag = np.where(single, b.flag & 0xF14, b.flag).astype(np.uint16)
    for i in np.nonzero(rng.random(b.n) < 0.03)[0]:
        b.qual[b.seq_off[i]:b.seq_off[i + 1]] = 0xFF
    return b


def indel_shape_cases(mref):
    from mapdamage_amd.batch import batch_from_records
    from tests.test_gpu_parity import _indel_records
    save_case("indelshapes_L70_A10_Q0", mref, batch_from_records(_indel_records(mref, 2500, 901)), LIBS2, 70, 10, 0)
    save_case("indelshapes_L70_A10_Q20", mref,
              batch_from_records(_indel_records(mref, 2500, 902, with_qual=True), with_qual=True), LIBS2, 70, 10, 20)
    save_case("indelshapes_L8_A3_Q20", mref,
              batch_from_records(_indel_records(mref, 1500, 903, with_qual=True), with_qual=True), LIBS2, 8, 3, 20)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only-indel-shapes", action="store_true")
    ap.add_argument("--time", action="store_true")
    args = ap.parse_args()
    if args.only_indel_shapes:
        indel_shape_cases(synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)),
                                            n_run=500, lower_run=3000))
        return

    # Appendix D hand vectors, tiny windows, with per-read gapped strings
    ref, batch = appendix_d_case()
    save_case("appendixD_L8_A3_Q0", ref, batch, LIBS1, 8, 3, 0, per_read=True)
    save_case("appendixD_L8_A3_Q20", ref, batch, LIBS1, 8, 3, 20, per_read=True)

    # edge set on the small genome, small and default windows, with/without -Q
    sref = synth.small_genome()
    edge = synth.make_edge_reads(sref, with_qual=True, nlib=2)
    save_case("edge_L8_A3_Q0", sref, edge, LIBS2, 8, 3, 0, per_read=True)
    save_case("edge_L8_A3_Q25", sref, edge, LIBS2, 8, 3, 25, per_read=True)
    save_case("edge_L70_A10_Q0", sref, edge, LIBS2, 70, 10, 0)
    save_case("edge_L1_A0_Q0", sref, edge, LIBS2, 1, 0, 0)
    save_case("edge_L200_A25_Q10", sref, edge, LIBS2, 200, 25, 10)

    # config 1 (BASELINE configs[0]): ~1k mixed reads + edge set, two libraries
    sref, c1 = synth.config1_batch()
    save_case("config1_L70_A10_Q0", sref, c1, LIBS2, 70, 10, 0)
    save_case("config1_L70_A10_Q20", sref, c1, LIBS2, 70, 10, 20)
    merged = c1.slice(0, c1.n)
    merged.lib[:] = 0
    save_case("config1_merged_L70_A10_Q0", sref, merged, MERGED, 70, 10, 0,
              note="--merge-libraries: every read maps to ('*','*') (reader.py:44-46)")
    same = c1.slice(0, c1.n)
    save_case("config1_samelib_L25_A5_Q0", sref, same, [("S", "L"), ("S", "L")], 25, 5, 0,
              note="two read groups naming the same (SM, LB): one library (reader.py:47-50)")

    # config 2/3/4-style samples on a mid-size genome
    mref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)),
                             n_run=500, lower_run=3000)
    save_case("config2s_L70_A10", mref, synth.config2_batch(mref, 4000, seed=2), LIBS1, 70, 10, 0)
    save_case("config3s_L70_A10", mref, synth.config3_batch(mref, 4000, seed=3), LIBS1, 70, 10, 0)
    save_case("config3s_L70_A10_Q15", mref, synth.config3_batch(mref, 3000, seed=33, with_qual=True),
              LIBS1, 70, 10, 15)
    save_case("config4s_L70_A10", mref, synth.config4_batch(mref, 4000, seed=4), LIBS1, 70, 10, 0)

    # single-indel shapes (indels of 1..130 bases anywhere in the read, two indels, indel + N, clips), with and
    # without qualities: the shapes the fast path splits into near / far entries, and the mask-by-left-index quirk
    # of align.py:65-71 behind an N operation
    indel_shape_cases(mref)

    # dnacomp_genome.csv through the reference's composition.write_base_comp, with its own native
    # seqtk extension compiled into oracle/_ref (never copied): SURVEY §8f N4
    import tempfile
    from mapdamage_amd import fasta
    from oracle import ref_seqtk
    seqtk = ref_seqtk.load()
    md = ref_harness.import_reference()
    sys.modules["mapdamage.seqtk"] = seqtk
    import importlib
    import mapdamage.composition as refcomp
    refcomp.seqtk = seqtk
    with tempfile.TemporaryDirectory() as tmp:
        fa = pathlib.Path(tmp) / "g.fa"
        fasta.write_fasta(fa, mref)
        refcomp.write_base_comp(fa, pathlib.Path(tmp) / "dnacomp_genome.csv")
        text = (pathlib.Path(tmp) / "dnacomp_genome.csv").read_bytes()
        per_contig = [[c["A"], c["C"], c["G"], c["T"]] for c in seqtk.comp(str(fa))]
    np.savez_compressed(GOLDEN / "genome_composition.npz",
                        ref_bases=np.frombuffer(b"".join(mref.seqs), dtype=np.uint8),
                        ref_lengths=np.asarray(mref.lengths, dtype=np.int64),
                        names=np.frombuffer(json.dumps(mref.names).encode(), dtype=np.uint8),
                        counts=np.asarray(per_contig, dtype=np.uint64),
                        csv=np.frombuffer(text, dtype=np.uint8))
    print("genome_composition         ", per_contig, text)

    # quality rescaling through the reference's own _rescale_qual_core (routing + per-read rescale)
    rref = synth.small_genome()
    rb = rescale_batch(rref)
    csv_text = rescale_csv()
    quals, mrs, log = ref_harness.run_reference_rescale(rref, rb, csv_text, 12, 10)
    qflat = rb.qual.copy()
    for i, q in enumerate(quals):
        if q is not None:
            s0, s1 = int(rb.seq_off[i]), int(rb.seq_off[i + 1])
            assert len(q) == s1 - s0
            qflat[s0:s1] = np.asarray(q, dtype=np.uint8)
    np.savez_compressed(GOLDEN / "genome_rescale.npz",
                        ref_bases=np.frombuffer(b"".join(rref.seqs), dtype=np.uint8),
                        ref_lengths=np.asarray(rref.lengths, dtype=np.int64),
                        names=np.frombuffer(json.dumps(rref.names).encode(), dtype=np.uint8),
                        flag=rb.flag, tid=rb.tid, pos=rb.pos, tlen=rb.tlen, cigar_off=rb.cigar_off, cigar=rb.cigar,
                        seq_off=rb.seq_off, seq=rb.seq, qual=rb.qual, mtid=rb.mtid, mpos=rb.mpos,
                        csv=np.frombuffer(csv_text.encode(), dtype=np.uint8), len5p=12, len3p=10,
                        qual_out=qflat, mr=np.asarray([np.nan if m is None else m for m in mrs], dtype=np.float64),
                        log=np.frombuffer(json.dumps(log).encode(), dtype=np.uint8))
    print("genome_rescale              reads=%d rescaled=%d" % (rb.n, sum(m is not None for m in mrs)))

    if args.time:
        big = synth.config2_batch(mref, 100_000, seed=2)
        t0 = time.perf_counter()
        res = ref_harness.run_reference(mref, big, LIBS1, 70, 10, 0)
        dt = time.perf_counter() - t0
        print("reference python path: %d reads in %.2fs = %.0f reads/s (1 core)" %
              (res["n_kept"], dt, res["n_kept"] / dt))


if __name__ == "__main__":
    main()
