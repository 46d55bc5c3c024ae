"""Host-side mirror of the reference's accumulator interface (mapdamage/statistics.py).

Same class names, same nested ``.data`` key space and the same ``write()`` output as the
reference, but filled from the dense device tables (``TableSet``) instead of a per-read Python
loop: ``MisincorporationRates`` (statistics.py:9-55), ``DNAComposition`` (:58-103),
``FragmentLengths`` (:106-137), ``check_table_and_warn_if_dmg_freq_is_low`` (:140-184)."""

import collections
import csv
import logging
import os

from . import layout as L
from .tables import TableSet


def _write_freq_table(table, columns, out, offset=0):
    """Sorted TSV emit with the derived ``Total`` column (format: SURVEY.md Appendix B)."""
    out.write("Sample\tLibrary\tEnd\tStd\tPos\t%s\n" % "\t".join(columns))
    for (sample, library) in sorted(table):
        ends = table[(sample, library)]
        for end in sorted(ends):
            for strand in sorted(ends[end]):
                sub = ends[end][strand]
                for index in sorted(sub[columns[0]]):
                    total = sum(sub[letter][index] for letter in L.LETTERS)
                    cells = [str(total) if col == "Total" else str(sub[col][index]) for col in columns]
                    out.write("\t".join([sample, library, end, strand, str(index + offset)] + cells))
                    out.write("\n")


class MisincorporationRates:
    def __init__(self, libraries, length):
        self.length = length
        self.data = {
            tuple(lib): {end: {strand: {col: dict.fromkeys(range(length), 0) for col in L.MIS_COLS}
                               for strand in L.STRANDS} for end in ("5p", "3p")}
            for lib in libraries}

    @classmethod
    def from_tables(cls, ts: TableSet):
        self = cls(ts.libraries, ts.length)
        for li, lib in enumerate(ts.libraries):
            for ei, end in enumerate(L.ENDS):
                for si, strand in enumerate(L.STRANDS):
                    block = ts.mis[li, ei, si]
                    for ci, col in enumerate(L.MIS_COLS):
                        self.data[tuple(lib)][end][strand][col] = {p: int(block[p, ci]) for p in range(ts.length)}
        return self

    def write(self, filepath):
        with open(filepath, "wt") as handle:
            _write_freq_table(self.data, L.MIS_HEADER, handle, offset=1)


class DNAComposition:
    def __init__(self, libraries, around, length):
        self.around, self.length = around, length
        keys = {"3p": L.comp_positions(0, length, around), "5p": L.comp_positions(1, length, around)}
        self.data = {
            tuple(lib): {end: {strand: {nt: dict.fromkeys(keys[end], 0) for nt in L.LETTERS}
                               for strand in L.STRANDS} for end in ("5p", "3p")}
            for lib in libraries}

    @classmethod
    def from_tables(cls, ts: TableSet):
        self = cls(ts.libraries, ts.around, ts.length)
        for li, lib in enumerate(ts.libraries):
            for ei, end in enumerate(L.ENDS):
                keys = L.comp_positions(ei, ts.length, ts.around)
                for si, strand in enumerate(L.STRANDS):
                    block = ts.comp[li, ei, si]
                    for bi, nt in enumerate(L.LETTERS):
                        self.data[tuple(lib)][end][strand][nt] = {k: int(block[ri, bi]) for ri, k in enumerate(keys)}
        return self

    def write(self, filepath):
        with open(filepath, "wt") as handle:
            _write_freq_table(self.data, L.COMP_HEADER, handle)


class FragmentLengths:
    def __init__(self, libraries):
        self.data = {tuple(lib): {(kind, strand): collections.defaultdict(int)
                                  for kind in ("pe", "se") for strand in L.STRANDS}
                     for lib in libraries}

    @classmethod
    def from_tables(cls, ts: TableSet):
        self = cls(ts.libraries)
        for li, k, s, ln, cnt in ts.lgd_sparse():
            self.data[tuple(ts.libraries[li])][(L.KINDS[k], L.STRANDS[s])][ln] += cnt
        return self

    def write(self, filepath):
        with open(filepath, "wt") as handle:
            handle.write("Sample\tLibrary\tStd\tKind\tLength\tOccurences\n")
            for (sample, library) in sorted(self.data):
                reads = self.data[(sample, library)]
                for (kind, strand) in sorted(reads):
                    for length in sorted(reads[(kind, strand)]):
                        handle.write("%s\t%s\t%s\t%s\t%d\t%d\n"
                                     % (sample, library, strand, kind, length, reads[(kind, strand)][length]))


def _position_one_totals(path):
    """Sums of the C / C>T columns of the 5p rows and of the G / G>A columns of the 3p rows at position 1 of a
    misincorporation table, over all samples, libraries and strands.  Raises KeyError / ValueError / OSError on a
    table that cannot be read that way; returns None for an empty file."""
    wanted = {"5p": ("C", "C>T"), "3p": ("G", "G>A")}
    totals = {("5p", "C"): 0, ("5p", "C>T"): 0, ("3p", "G"): 0, ("3p", "G>A"): 0}
    with open(path, newline="") as handle:
        lines = (line.rstrip("\r\n") for line in handle)
        header = next(lines, None)
        if not header:
            return None
        column = {name: i for i, name in enumerate(header.split("\t"))}
        for line in lines:
            fields = line.split("\t")
            if int(fields[column["Pos"]]) != 1:
                continue
            end = fields[column["End"]]
            for name in wanted[end]:
                totals[(end, name)] += int(fields[column[name]])
    return totals


def check_table_and_warn_if_dmg_freq_is_low(folder):
    """The reference's check of the same name (used by its main flow right after the tables are written): False when
    `misincorporation.txt` is unusable, True otherwise, with a warning when the first-position damage — 5p C>T plus
    3p G>A frequency — is below 1 %.  Same messages as the reference."""
    logger = logging.getLogger(__name__)
    filename = "misincorporation.txt"
    try:
        totals = _position_one_totals(os.path.join(folder, filename))
    except (OSError, KeyError, ValueError, IndexError) as error:
        logger.error("Error reading misincorporation table: %s", error)
        return False
    if totals is None:
        logger.error("%r is empty; please re-run mapDamage", filename)
        return False
    c_sites, g_sites = totals[("5p", "C")], totals[("3p", "G")]
    if c_sites == 0 or g_sites == 0:
        logger.error("Insufficient data in %r; cannot perform Bayesian computation", filename)
        return False
    damage = totals[("5p", "C>T")] / c_sites + totals[("3p", "G>A")] / g_sites
    if damage < 0.01:
        logger.warning("DNA damage levels are too low, the Bayesian computation should not be "
                       "performed (%f < 0.01)", damage)
    return True
