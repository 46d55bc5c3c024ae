"""Shared helpers: golden fixture loading and table comparison."""

import json
import pathlib

import numpy as np

from mapdamage_amd.batch import ReadBatch, Reference
from mapdamage_amd.tables import TableSet, merge_library_ids

GOLDEN = pathlib.Path(__file__).resolve().parent / "golden"


# fixtures of other kinds in the same folder (no count tables: the reference's downsampling draws, the hand-assembled BAM's truth)
_NOT_TABLES = {"downsample", "foreign_bam"}


def golden_names():
    return sorted(p.stem for p in GOLDEN.glob("*.npz") if not p.stem.startswith("genome_") and p.stem not in _NOT_TABLES)


class Golden:
    def __init__(self, name):
        z = np.load(GOLDEN / (name + ".npz"))
        self.meta = json.loads(bytes(z["meta"]).decode())
        lens = z["ref_lengths"]
        bases = bytes(z["ref_bases"])
        seqs, o = [], 0
        for ln in lens:
            seqs.append(bases[o:o + int(ln)])
            o += int(ln)
        self.ref = Reference(list(self.meta["contig_names"]), seqs)
        qual = z["qual"] if "qual" in z.files else None
        self.batch = ReadBatch(z["flag"], z["lib"], z["tid"], z["pos"], z["tlen"], z["cigar_off"],
                               z["cigar"], z["seq_off"], z["seq"], qual).validate()
        self.length = self.meta["length"]
        self.around = self.meta["around"]
        self.minqual = self.meta["minqual"]
        # read-group order -> unique library ids (what the host does before packing)
        self.libraries, remap = merge_library_ids([tuple(x) for x in self.meta["libraries"]])
        self.batch.lib = remap[self.batch.lib]
        self.sorted_libraries = [tuple(x) for x in self.meta["sorted_libraries"]]
        self.mis = z["mis"]          # sorted-library order
        self.comp = z["comp"]
        self.lgd = z["lgd"]          # rows (sorted lib index, kind, strand, len, count)
        self.txt = {"misincorporation.txt": bytes(z["txt_mis"]).decode(),
                    "dnacomp.txt": bytes(z["txt_comp"]).decode(),
                    "lgdistribution.txt": bytes(z["txt_lgd"]).decode()}
        self.n_kept = self.meta["n_kept"]
        self.per_read = json.loads(bytes(z["per_read"]).decode()) if "per_read" in z.files else None

    def check(self, ts: TableSet):
        """Assert a TableSet (library-id order) equals the golden tables and texts."""
        order = [self.libraries.index(lib) for lib in self.sorted_libraries]
        np.testing.assert_array_equal(ts.mis[order], self.mis)
        np.testing.assert_array_equal(ts.comp[order], self.comp)
        got = [(order.index(li), k, s, ln, c) for li, k, s, ln, c in ts.lgd_sparse()]
        assert sorted(got) == [tuple(int(v) for v in row) for row in self.lgd]
        assert ts.misincorporation_text() == self.txt["misincorporation.txt"]
        assert ts.dnacomp_text() == self.txt["dnacomp.txt"]
        assert ts.lgdistribution_text() == self.txt["lgdistribution.txt"]
        assert ts.n_kept == self.n_kept


def oracle_tableset(ref, batch, libraries, length, around, minqual=0, lgd_max=65536):
    from oracle import oracle
    r = oracle.tabulate(ref, batch, len(libraries), length, around, minqual, lgd_max)
    return TableSet(list(libraries), length, around, r["mis"], r["comp"], r["lgd"], r["lgd_over"],
                    r["n_kept"])


def assert_tables_equal(a: TableSet, b: TableSet):
    np.testing.assert_array_equal(a.mis, b.mis)
    np.testing.assert_array_equal(a.comp, b.comp)
    assert a.lgd_sparse() == b.lgd_sparse()
    assert a.n_kept == b.n_kept
