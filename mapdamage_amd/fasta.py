"""FASTA / .fai readers and the BAM-vs-FASTA dictionary check (mapdamage/seq.py:38-112)."""

import gzip
import logging

from .batch import Reference


def read_fasta(path):
    """Whole FASTA (optionally gzip-compressed) -> ([names], [bytes]) with the original case."""
    opener = gzip.open if str(path).endswith(".gz") else open
    names, seqs, cur = [], [], []
    with opener(path, "rb") as handle:
        for line in handle:
            if line.startswith(b">"):
                if names:
                    seqs.append(b"".join(cur))
                header = line[1:].split()
                names.append(header[0].decode() if header else "")
                cur = []
            elif names:
                cur.append(line.strip())
    if names:
        seqs.append(b"".join(cur))
    return names, seqs


def write_fasta(path, ref: Reference, width=60):
    """FASTA + .fai (samtools faidx layout) of ``ref``; the line ends are put in by a reshape, not a loop over the lines
    (a genome of human size is fifty million of them)."""
    import numpy as np
    with open(path, "wb") as handle, open(str(path) + ".fai", "wt") as fai:
        offset = 0
        for name, seq in zip(ref.names, ref.seqs):
            header = (">%s\n" % name).encode()
            handle.write(header)
            offset += len(header)
            fai.write("%s\t%d\t%d\t%d\t%d\n" % (name, len(seq), offset, width, width + 1))
            whole = len(seq) // width * width
            if whole:
                lines = np.empty((whole // width, width + 1), np.uint8)
                lines[:, :width] = np.frombuffer(seq, np.uint8, whole).reshape(-1, width)
                lines[:, width] = 10
                handle.write(lines.data)
            if len(seq) > whole:
                handle.write(bytes(seq[whole:]) + b"\n")
            offset += len(seq) + (len(seq) + width - 1) // width


class FastaOnDisk:
    """The sequences of a BAM header (``tid`` order) as they lie in an indexed FASTA file: what ``pysam.FastaFile``
    (main.py:115) is to the reference's loop.  Nothing is read here: ``DamageEngine.set_reference`` hands the path to the
    library, which sends the file's bytes to HBM and strips the line ends there (include/mdx.h
    ``mdx_set_reference_fasta``); ``lengths`` are the index's until then, the loaded ones afterwards."""

    def __init__(self, path, names, lengths, missing_ok=False):
        self.path, self.names, self.lengths, self.missing_ok = str(path), list(names), list(lengths), bool(missing_ok)


def ensure_fasta_index(path):
    """``<path>.fai`` exists afterwards (htslib builds it when ``pysam.FastaFile`` opens a file without one, main.py:115);
    raises ValueError with the library's message when the file cannot be indexed."""
    import ctypes
    import os
    if os.path.exists(str(path) + ".fai"):
        return
    from .engine import load_library
    err = ctypes.create_string_buffer(512)
    if load_library().mdx_fasta_index(str(path).encode(), err, 512) != 0:
        raise ValueError("cannot index %r: %s" % (str(path), err.value.decode(errors="replace")))


class _FaiError(ValueError):
    """A malformed .fai line: carries the log message and its arguments."""

    def __init__(self, message, *args):
        super().__init__(message % args)
        self.message, self.args_ = message, args


def _fai_records(filename):
    """(name, length) per line of a samtools faidx index; raises _FaiError on the first malformed line."""
    with open(filename, "r") as handle:
        for lineno, line in enumerate(handle, 1):
            columns = line.split("\t")
            if len(columns) != 5:
                raise _FaiError("Line %i in %r contains wrong number of fields, found %i, expected 5:",
                                lineno, filename, len(columns))
            name, length = columns[0], columns[1]
            if not length.strip().lstrip("+-").isdigit():
                raise _FaiError("Length at line %i in %r is not a number; found %r", lineno, filename, length)
            yield name, int(length)


def read_fasta_index(filename):
    """.fai -> {name: length}, or None after logging what is wrong with it (the reference's
    seq.read_fasta_index contract: same messages, same None)."""
    logger = logging.getLogger(__name__)
    try:
        lengths = dict(_fai_records(filename))
    except _FaiError as error:
        logger.error(error.message, *error.args_)
        return None
    if lengths:
        return lengths
    for message, args in (("Error: Index for %r does contain any sequences.", (filename,)),
                          ("Please ensure that FASTA file is valid, and", ()),
                          ("re-index file using 'samtool faidx'.", ())):
        logger.error(message, *args)
    return None


def compare_sequence_dicts(fasta_dict, bam_dict):
    """Does the FASTA provide every sequence of the BAM header at the same length?  Logs the reference's
    messages (seq.compare_sequence_dicts): length mismatches and sequences missing from the FASTA are errors,
    sequences only the FASTA has a warning."""
    if fasta_dict == bam_dict:
        return True
    logger = logging.getLogger(__name__)
    shared = fasta_dict.keys() & bam_dict.keys()
    if not shared:
        logger.error("BAM and FASTA file have no sequence names in common")
        return False
    problems = {
        "mismatched": [(name, fasta_dict[name], bam_dict[name]) for name in sorted(shared)
                       if fasta_dict[name] != bam_dict[name]],
        "missing": [(name, bam_dict[name]) for name in bam_dict.keys() - shared],
        "extra": [(name, fasta_dict[name]) for name in fasta_dict.keys() - shared],
    }
    reports = (("mismatched", logger.error, "Length of required FASTA sequences differ:", " - %s: %i vs %i bp"),
               ("missing", logger.error, "Sequences not found in FASTA:", "%s (%i bp)"),
               ("extra", logger.warning, "FASTA file contains extra sequences:", " - %s = %i bp"))
    for kind, log, title, row in reports:
        if problems[kind]:
            log(title)
            for values in problems[kind]:
                log(row % values)
    return not (problems["mismatched"] or problems["missing"])


def reference_for_bam(fasta_path, bam_names, missing_ok=False):
    """Contigs of the FASTA reordered to BAM ``tid`` order (chrom lookup is by name,
    main.py:175-180).  ``missing_ok``: a sequence the FASTA lacks becomes an empty contig — a record that maps to it
    is then a bad record when it is met (the reference fails in ``fetch`` at that read, not before).
    An uncompressed FASTA stays on disk (``FastaOnDisk``: the library loads it, no pass over the bases here); a
    gzip-compressed one is read here."""
    if not str(fasta_path).endswith(".gz"):
        ensure_fasta_index(fasta_path)
        have = dict(_fai_records(str(fasta_path) + ".fai"))
        if not missing_ok:
            for n in bam_names:
                if n not in have:
                    raise KeyError(n)
        return FastaOnDisk(fasta_path, bam_names, [have.get(n, 0) for n in bam_names], missing_ok)
    return reference_in_memory(fasta_path, bam_names, missing_ok)


def reference_in_memory(fasta_path, bam_names, missing_ok=False):
    """The same contigs read into host memory by ``read_fasta`` (gzip-compressed files; the tests' second opinion on
    the library's loader)."""
    names, seqs = read_fasta(fasta_path)
    by_name = dict(zip(names, seqs))
    return Reference(list(bam_names), [by_name.get(n, b"") if missing_ok else by_name[n] for n in bam_names])
