"""Genome base composition -> ``dnacomp_genome.csv`` (mirror of mapdamage/composition.py).

The reference counts bases with its native ``seqtk`` extension (seqtk.c:79-104); here the counts
come from the reference already resident on the GPU (``DamageEngine.genome_composition``).  The CSV
is written exactly as the reference does: ``csv.writer`` defaults (``\\r\\n`` line ends), header
``A,C,G,T``, one row of ``count / total`` floats in Python ``repr`` form."""

import csv

_BASES = ("A", "C", "G", "T")


def base_frequencies(counts):
    """Per-contig counts [n_contig][4] (A, C, G, T; ``DamageEngine.genome_composition``) -> the genome-wide frequency
    of each base (what mapdamage/composition.py:10-17 derives from seqtk's counts)."""
    totals = [sum(int(row[k]) for row in counts) for k in range(len(_BASES))]
    n = sum(totals)
    return {base: total / n for base, total in zip(_BASES, totals)}


def write_base_comp(counts, destination):
    """``dnacomp_genome.csv``: a header row and one row of frequencies, csv-module formatting (``repr`` floats, CRLF)."""
    freqs = base_frequencies(counts)
    with open(destination, "wt", newline="") as handle:
        csv.writer(handle).writerows([_BASES, [freqs[base] for base in _BASES]])


def read_base_comp(filename):
    """The frequencies of a file written by ``write_base_comp`` as ``{base: text}`` (mapdamage/composition.py:28-35)."""
    with open(filename, newline="") as handle:
        rows = csv.reader(handle)
        header, first = next(rows, None), next(rows, None)
    if not header or first is None:
        raise csv.Error("No rows found in %r" % (filename,))
    return dict(zip(header, first))
