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
    with open(path, "wb") as handle, open(str(path) + ".fai", "wt") as fai:
        offset = 0
        for name, seq in zip(ref.names, ref.seqs):
            header = (">%s\n" % name).encode()
            handle.write(header)
            offset += len(header)
            fai.write("%s\t%d\t%d\t%d\t%d\n" % (name, len(seq), offset, width, width + 1))
            for i in range(0, len(seq), width):
                handle.write(seq[i:i + width] + b"\n")
            offset += len(seq) + (len(seq) + width - 1) // width


def read_fasta_index(filename):
    """.fai -> {name: length}; None (with logged errors) when malformed (seq.py:38-72)."""
    logger = logging.getLogger(__name__)
    fai = {}
    with open(filename, "r") as handle:
        for lineno, line in enumerate(handle, 1):
            fields = line.split("\t")
            if len(fields) != 5:
                logger.error("Line %i in %r contains wrong number of fields, found %i, expected 5:",
                             lineno, filename, len(fields))
                return None
            try:
                fai[fields[0]] = int(fields[1])
            except ValueError:
                logger.error("Length at line %i in %r is not a number; found %r", lineno, filename, fields[1])
                return None
    if not fai:
        logger.error("Error: Index for %r does contain any sequences.", filename)
        logger.error("Please ensure that FASTA file is valid, and")
        logger.error("re-index file using 'samtool faidx'.")
        return None
    return fai


def compare_sequence_dicts(fasta_dict, bam_dict):
    """True when every BAM sequence exists in the FASTA with the same length (seq.py:75-112)."""
    if fasta_dict == bam_dict:
        return True
    logger = logging.getLogger(__name__)
    common = set(fasta_dict) & set(bam_dict)
    if not common:
        logger.error("BAM and FASTA file have no sequence names in common")
        return False
    different = [(k, fasta_dict[k], bam_dict[k]) for k in sorted(common) if fasta_dict[k] != bam_dict[k]]
    if different:
        logger.error("Length of required FASTA sequences differ:")
        for values in different:
            logger.error(" - %s: %i vs %i bp" % values)
    bam_only = set(bam_dict) - common
    if bam_only:
        logger.error("Sequences not found in FASTA:")
        for key in bam_only:
            logger.error("%s (%i bp)", key, bam_dict[key])
    fasta_only = set(fasta_dict) - common
    if fasta_only:
        logger.warning("FASTA file contains extra sequences:")
        for key in fasta_only:
            logger.warning(" - %s = %i bp", key, fasta_dict[key])
    return not (different or bam_only)


def reference_for_bam(fasta_path, bam_names):
    """Contigs of the FASTA reordered to BAM ``tid`` order (chrom lookup is by name,
    main.py:175-180)."""
    names, seqs = read_fasta(fasta_path)
    by_name = dict(zip(names, seqs))
    return Reference(list(bam_names), [by_name[n] for n in bam_names])
