"""Genome base composition -> ``dnacomp_genome.csv`` (mirror of mapdamage/composition.py).

The reference counts bases with its native ``seqtk`` extension (seqtk.c:79-104); here the counts
come from the reference already resident on the GPU (``DamageEngine.genome_composition``).  The CSV
is written exactly as the reference does: ``csv.writer`` defaults (``\\r\\n`` line ends), header
``A,C,G,T``, one row of ``count / total`` floats in Python ``repr`` form."""

import csv


def base_frequencies(counts):
    """counts: array [n_contig][4] (A, C, G, T) -> dict of frequencies (composition.py:10-17)."""
    bases = {"A": 0, "C": 0, "G": 0, "T": 0}
    for row in counts:
        for key, value in zip("ACGT", row):
            bases[key] += int(value)
    total = sum(bases.values())
    return {key: bases[key] / total for key in bases}


def write_base_comp(counts, destination):
    freqs = base_frequencies(counts)
    with open(destination, "wt", newline="") as handle:
        writer = csv.writer(handle)
        header = ["A", "C", "G", "T"]
        writer.writerow(header)
        writer.writerow(freqs[key] for key in header)


def read_base_comp(filename):
    """First data row of a file written by ``write_base_comp`` (composition.py:28-35)."""
    with open(filename, newline="") as handle:
        for row in csv.DictReader(handle):
            return row
    raise csv.Error("No rows found in %r" % (filename,))
