"""Genome base composition / dnacomp_genome.csv.  The golden comes from the
reference's composition.write_base_comp over its own native seqtk extension (oracle/_ref)."""

import json
import pathlib

import numpy as np
import pytest

from mapdamage_amd import composition, fasta
from mapdamage_amd.batch import Reference

GOLDEN = pathlib.Path(__file__).resolve().parent / "golden" / "genome_composition.npz"


def load():
    z = np.load(GOLDEN)
    names = json.loads(bytes(z["names"]).decode())
    seqs, o = [], 0
    bases = bytes(z["ref_bases"])
    for ln in z["ref_lengths"]:
        seqs.append(bases[o:o + int(ln)])
        o += int(ln)
    return Reference(names, seqs), z["counts"], bytes(z["csv"])


def numpy_counts(ref):
    out = []
    for s in ref.seqs:
        a = np.frombuffer(s, dtype=np.uint8) & 0xDF
        out.append([int((a == ord(c)).sum()) for c in "ACGT"])
    return np.asarray(out, dtype=np.uint64)


def test_counts_and_csv_match_reference_golden(tmp_path):
    ref, counts, csv_bytes = load()
    np.testing.assert_array_equal(numpy_counts(ref), counts)
    composition.write_base_comp(counts, tmp_path / "dnacomp_genome.csv")
    assert (tmp_path / "dnacomp_genome.csv").read_bytes() == csv_bytes
    row = composition.read_base_comp(tmp_path / "dnacomp_genome.csv")
    assert set(row) == {"A", "C", "G", "T"} and abs(sum(float(v) for v in row.values()) - 1) < 1e-12


def test_reference_native_extension_agrees(tmp_path):
    """oracle/_ref/seqtk*.so is the reference's seqtk.c compiled as is."""
    from oracle import ref_seqtk
    seqtk = ref_seqtk.load(build=True)
    if seqtk is None:
        pytest.skip("oracle/_ref not built and /root/reference absent")
    ref, counts, _ = load()
    fasta.write_fasta(tmp_path / "g.fa", ref)
    got = [[c["A"], c["C"], c["G"], c["T"]] for c in seqtk.comp(str(tmp_path / "g.fa"))]
    np.testing.assert_array_equal(np.asarray(got, dtype=np.uint64), counts)


@pytest.mark.gpu
def test_hip_genome_composition_matches_golden(tmp_path):
    from mapdamage_amd.engine import DamageEngine
    ref, counts, csv_bytes = load()
    with DamageEngine([("s", "l")]) as eng:
        eng.set_reference(ref)
        got = eng.genome_composition(len(ref.names))
    np.testing.assert_array_equal(got, counts)
    composition.write_base_comp(got, tmp_path / "out.csv")
    assert (tmp_path / "out.csv").read_bytes() == csv_bytes
