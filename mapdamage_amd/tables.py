"""Dense count tables and their byte-exact text emitters.

``TableSet`` is what ``DamageEngine.finish()`` returns: the canonical dense uint64 tables of
``layout.py`` indexed by library *id*, plus the emitters that reproduce the reference's
``misincorporation.txt`` / ``dnacomp.txt`` / ``lgdistribution.txt`` byte for byte
(mapdamage/statistics.py:53-55,95-98,128-137,187-203; format in SURVEY.md Appendix B).
"""

import io
from dataclasses import dataclass, field

import numpy as np

from . import layout as L


@dataclass
class TableSet:
    libraries: list            # [(sample, library)] indexed by library id
    length: int
    around: int
    mis: np.ndarray            # u64 [nlib][2][2][L][25]
    comp: np.ndarray           # u64 [nlib][2][2][L+A][4]
    lgd: np.ndarray            # u64 [nlib][2][2][lgd_max]
    lgd_over: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.int64))
    n_kept: int = 0

    # ------------------------------------------------------------------ helpers
    def _sorted_libs(self):
        """(lib tuple, id) in the order of ``sorted(table.items())`` (statistics.py:190)."""
        return sorted((tuple(lib), i) for i, lib in enumerate(self.libraries))

    def lgd_sparse(self):
        """Sorted rows (lib id, kind, strand, length, count) of the length histogram."""
        idx = np.argwhere(self.lgd > 0)
        rows = {}
        for li, k, s, ln in idx:
            rows[(int(li), int(k), int(s), int(ln))] = int(self.lgd[li, k, s, ln])
        for li, k, s, ln in self.lgd_over:
            key = (int(li), int(k), int(s), int(ln))
            rows[key] = rows.get(key, 0) + 1
        return sorted(key + (cnt,) for key, cnt in rows.items())

    def add(self, other):
        assert self.mis.shape == other.mis.shape and self.comp.shape == other.comp.shape
        self.mis += other.mis
        self.comp += other.comp
        self.lgd += other.lgd
        self.lgd_over = np.concatenate([self.lgd_over, other.lgd_over])
        self.n_kept += other.n_kept
        return self

    # ------------------------------------------------------------------ emitters
    def misincorporation_text(self):
        out = io.StringIO()
        out.write("Sample\tLibrary\tEnd\tStd\tPos\t%s\n" % "\t".join(L.MIS_HEADER))
        for (sample, library), li in self._sorted_libs():
            for ei, end in enumerate(L.ENDS):
                for si, strand in enumerate(L.STRANDS):
                    block = self.mis[li, ei, si]
                    totals = block[:, :4].sum(axis=1)
                    for p in range(self.length):
                        row = block[p]
                        cells = [str(int(x)) for x in row[:4]]
                        cells.append(str(int(totals[p])))
                        cells.extend(str(int(x)) for x in row[4:])
                        out.write("%s\t%s\t%s\t%s\t%d\t%s\n"
                                  % (sample, library, end, strand, p + 1, "\t".join(cells)))
        return out.getvalue()

    def dnacomp_text(self):
        out = io.StringIO()
        out.write("Sample\tLibrary\tEnd\tStd\tPos\t%s\n" % "\t".join(L.COMP_HEADER))
        for (sample, library), li in self._sorted_libs():
            for ei, end in enumerate(L.ENDS):
                keys = L.comp_positions(ei, self.length, self.around)
                for si, strand in enumerate(L.STRANDS):
                    block = self.comp[li, ei, si]
                    totals = block.sum(axis=1)
                    for ri, key in enumerate(keys):
                        row = block[ri]
                        out.write("%s\t%s\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\n"
                                  % (sample, library, end, strand, key, int(row[0]), int(row[1]),
                                     int(row[2]), int(row[3]), int(totals[ri])))
        return out.getvalue()

    def lgdistribution_text(self):
        out = io.StringIO()
        out.write("Sample\tLibrary\tStd\tKind\tLength\tOccurences\n")
        sparse = self.lgd_sparse()
        by_lib = {}
        for li, k, s, ln, cnt in sparse:
            by_lib.setdefault(li, []).append((k, s, ln, cnt))
        for (sample, library), li in self._sorted_libs():
            for k, s, ln, cnt in sorted(by_lib.get(li, [])):
                out.write("%s\t%s\t%s\t%s\t%d\t%d\n"
                          % (sample, library, L.STRANDS[s], L.KINDS[k], ln, cnt))
        return out.getvalue()

    def damage_frequency_text(self, end, readplot):
        """EXPERIMENTAL, parity unpinned: ``5pCtoT_freq.txt`` (end "5p") or
        ``3pGtoA_freq.txt`` (end "3p") as mapDamage 2.0-2.2 wrote them from R; the reference snapshot
        no longer produces these files, so the format is recalled, not checked.  Per position 1..readplot:
        sum over libraries and strands of C>T / C at the 5' end (G>A / G at the 3' end), the aggregation of
        ``calculate.mutation.table`` (mapdamage/r/mapDamage.r:81-92)."""
        ei = L.ENDS.index(end)
        num_col, den_col, name = ("C>T", "C", "5pC>T") if end == "5p" else ("G>A", "G", "3pG>A")
        num = self.mis[:, ei, :, :, L.MIS_COLS.index(num_col)].sum(axis=(0, 1))
        den = self.mis[:, ei, :, :, L.MIS_COLS.index(den_col)].sum(axis=(0, 1))
        out = io.StringIO()
        out.write("pos\t%s\n" % name)
        for p in range(min(readplot, self.length)):
            freq = float(num[p]) / float(den[p]) if den[p] else float("nan")
            out.write("%d\t%s\n" % (p + 1, "NaN" if freq != freq else "%.15g" % freq))
        return out.getvalue()

    def write(self, folder):
        """Write the three tables into ``folder`` (mapdamage/main.py:229-231)."""
        import pathlib
        folder = pathlib.Path(folder)
        (folder / "misincorporation.txt").write_text(self.misincorporation_text())
        (folder / "dnacomp.txt").write_text(self.dnacomp_text())
        (folder / "lgdistribution.txt").write_text(self.lgdistribution_text())


def table_words(nlib, length, around, lgd_max):
    """uint64 words of the packed table block (include/mdx.h: mdx_table_words)."""
    return nlib * 4 * length * L.N_MIS_COLS + nlib * 4 * (length + around) * 4 + nlib * 4 * lgd_max + 2


def pack_words(ts: "TableSet") -> np.ndarray:
    """TableSet -> packed block [mis | comp | lgd | n_kept | n_lgd_over] (the all-reduce message)."""
    tail = np.asarray([ts.n_kept, ts.lgd_over.shape[0]], dtype=np.uint64)
    return np.concatenate([ts.mis.reshape(-1), ts.comp.reshape(-1), ts.lgd.reshape(-1), tail]).astype(np.uint64)


def unpack_words(words, libraries, length, around, lgd_max, lgd_over=None) -> "TableSet":
    """Packed block (host copy of mdx_finish_device output, possibly all-reduced) -> TableSet."""
    nlib = len(libraries)
    nm = nlib * 4 * length * L.N_MIS_COLS
    nc = nlib * 4 * (length + around) * 4
    nl = nlib * 4 * lgd_max
    words = np.ascontiguousarray(words).view(np.uint64)
    assert words.shape[0] == nm + nc + nl + 2, (words.shape, nm + nc + nl + 2)
    mis = words[:nm].reshape(nlib, 2, 2, length, L.N_MIS_COLS).copy()
    comp = words[nm:nm + nc].reshape(nlib, 2, 2, length + around, 4).copy()
    lgd = words[nm + nc:nm + nc + nl].reshape(nlib, 2, 2, lgd_max).copy()
    over = np.zeros((0, 4), np.int64) if lgd_over is None else np.asarray(lgd_over, np.int64).reshape(-1, 4)
    return TableSet([tuple(x) for x in libraries], length, around, mis, comp, lgd, over, int(words[-2]))


def merge_library_ids(libraries):
    """Map read-group order to unique library ids (reader.py:47-50: several read groups may
    name the same (SM, LB)).  Returns (unique list, remap array old id -> new id)."""
    uniq, remap = [], []
    index = {}
    for lib in libraries:
        lib = tuple(lib)
        if lib not in index:
            index[lib] = len(uniq)
            uniq.append(lib)
        remap.append(index[lib])
    return uniq, np.asarray(remap, dtype=np.uint16)
