"""Drive the *reference's own* hot-path functions (build container only).

This module imports ``/root/reference/mapdamage`` (with empty stand-in modules for the
two third-party imports it cannot satisfy here) and replays the loop
body of mapdamage/main.py:165-217 over duck-typed read objects built from a
``ReadBatch``.  It exists to *generate golden vectors* (``tools/make_golden.py``) and to
time the reference's Python path; it never ships to the GPU box and nothing under
``tests/``, ``bench.py`` or the product imports it.

pysam attribute semantics emulated by ``_Read`` (third-party and
therefore "parity unpinned" at that boundary):

* ``query``/``qqual``: SEQ/QUAL with leading and trailing soft clips removed
* ``aend``: htslib ``bam_endpos`` = pos + max(1, reference-consuming length); ``None``
  when the record has no CIGAR
* ``qual``: ``None`` when the BAM record has no qualities (first byte 0xFF)
"""

import io
import pathlib
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def import_reference():
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.modules.setdefault("pysam", types.ModuleType("pysam"))
    sys.modules.setdefault("coloredlogs", types.ModuleType("coloredlogs"))
    import mapdamage
    import mapdamage.statistics
    import mapdamage.reader
    return mapdamage


class _Read:
    __slots__ = ("flag", "tid", "pos", "aend", "cigar", "query", "qual", "qqual",
                 "template_length", "reference_length", "is_reverse", "is_paired",
                 "is_proper_pair", "is_read1", "lib")

    def __init__(self, rec):
        self.flag = rec["flag"]
        self.tid = rec["tid"]
        self.pos = rec["pos"]
        self.cigar = [(op, ln) for op, ln in rec["cigar"]]
        self.template_length = rec["tlen"]
        self.is_reverse = bool(self.flag & 0x10)
        self.is_paired = bool(self.flag & 0x1)
        self.is_proper_pair = bool(self.flag & 0x2)
        self.is_read1 = bool(self.flag & 0x40)
        self.lib = rec["lib"]
        seq = rec["seq"]
        start = 0
        for op, ln in self.cigar:
            if op == 5:
                continue
            if op == 4:
                start += ln
            else:
                break
        end = len(seq)
        for op, ln in reversed(self.cigar[1:]):
            if op == 5:
                continue
            if op == 4:
                end -= ln
            else:
                break
        self.query = seq[start:end]
        if rec["qual"] is None:
            self.qual = None
            self.qqual = None
        else:
            self.qual = "".join(chr(q + 33) for q in rec["qual"])
            self.qqual = self.qual[start:end]
        if not self.cigar:
            self.aend = None
            self.reference_length = None
        else:
            rlen = sum(ln for op, ln in self.cigar if op in (0, 2, 3, 7, 8))
            self.aend = self.pos + (rlen if rlen else 1)
            self.reference_length = self.aend - self.pos


class _Fasta:
    def __init__(self, ref):
        self._seqs = {name: s.decode("latin-1") for name, s in zip(ref.names, ref.seqs)}

    def fetch(self, chrom, start, end):
        s = self._seqs[chrom]
        if start > end:
            raise ValueError("start > end")
        return s[start:min(end, len(s))]


def run_reference(ref, batch, libraries, length, around, minqual, per_read=False):
    """Replay main.py:165-217 in order.  ``libraries``: list of (SM, LB) tuples indexed
    by ``batch.lib``.  Returns dict(mis, comp, lgd, texts, n_kept[, per_read])."""
    md = import_reference()
    stats = md.statistics
    align = md.align
    revcomp = md.seq.revcomp
    filtered = md.reader.BAMReader._filter_reads

    fasta = _Fasta(ref)
    reflengths = dict(zip(ref.names, ref.lengths))
    libs = list(dict.fromkeys(libraries))
    misincorp = stats.MisincorporationRates(libs, length)
    dnacomp = stats.DNAComposition(libs, around, length)
    lgdistrib = stats.FragmentLengths(libs)

    reads = (_Read(batch.record(i)) for i in range(batch.n))
    trace = []
    counter = 0
    for read in filtered(reads):
        counter += 1
        library = libraries[read.lib]
        coordinate = align.get_coordinates(read)
        lgdistrib.update(read, library)
        chrom = ref.names[read.tid]
        before, after = align.get_around(coordinate, chrom, reflengths, around, fasta)
        refseq = fasta.fetch(chrom, min(coordinate), max(coordinate)).upper()
        seq = read.query
        if not (minqual and read.qual):
            seq, refseq = align.align(read.cigar, seq, refseq)
        else:
            seq, _, refseq = align.align_with_qual(read.cigar, seq, read.qqual, minqual, refseq)
        if read.is_reverse:
            refseq = revcomp(refseq)
            seq = revcomp(seq)
            beforerev = revcomp(after)
            after = revcomp(before)
            before = beforerev
        misincorp.update_soft_clipping(read, library)
        misincorp.update(read, seq, refseq, "5p", library)
        misincorp.update(read, reversed(seq), reversed(refseq), "3p", library)
        dnacomp.update_read(read, length, library)
        dnacomp.update_reference(read, before, after, library)
        if per_read:
            trace.append((seq, refseq, before, after))

    texts = {}
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        tmp = pathlib.Path(tmp)
        misincorp.write(tmp / "misincorporation.txt")
        dnacomp.write(tmp / "dnacomp.txt")
        lgdistrib.write(tmp / "lgdistribution.txt")
        for name in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt"):
            texts[name] = (tmp / name).read_text()

    out = dict(mis=misincorp.data, comp=dnacomp.data, lgd=lgdistrib.data, texts=texts,
               n_kept=counter)
    if per_read:
        out["per_read"] = trace
    return out


def dense_tables(result, libraries, length, around):
    """Nested reference dicts -> canonical dense arrays (layout.py docstring), with the
    libraries in *sorted* order (the order of statistics.py:190)."""
    from mapdamage_amd import layout as L
    libs = sorted(dict.fromkeys(libraries))
    mis = np.zeros((len(libs), 2, 2, length, L.N_MIS_COLS), np.uint64)
    comp = np.zeros((len(libs), 2, 2, length + around, 4), np.uint64)
    lgd = []
    for li, lib in enumerate(libs):
        for ei, end in enumerate(L.ENDS):
            keys = L.comp_positions(ei, length, around)
            for si, strand in enumerate(L.STRANDS):
                sub = result["mis"][lib][end][strand]
                for ci, col in enumerate(L.MIS_COLS):
                    for p in range(length):
                        mis[li, ei, si, p, ci] = sub[col][p]
                subc = result["comp"][lib][end][strand]
                for bi, base in enumerate(L.LETTERS):
                    for ri, key in enumerate(keys):
                        comp[li, ei, si, ri, bi] = subc[base][key]
        for (kind, strand), lengths in result["lgd"][lib].items():
            for ln, cnt in lengths.items():
                if cnt:
                    lgd.append((li, L.KINDS.index(kind), L.STRANDS.index(strand), ln, cnt))
    lgd = np.asarray(sorted(lgd), dtype=np.int64).reshape(-1, 5)
    return libs, mis, comp, lgd


# ------------------------------------------------------------------------------------ rescale
class _RescaleRead(_Read):
    """Adds what mapdamage/rescale.py touches: mate fields, tags, a settable ``qual``."""
    __slots__ = ("is_unmapped", "mate_is_reverse", "pnext", "mrnm", "qname", "tags", "index")

    def __init__(self, rec, index, mtid, mpos):
        super().__init__(rec)
        self.is_unmapped = bool(self.flag & 0x4)
        self.mate_is_reverse = bool(self.flag & 0x20)
        self.pnext = mpos
        self.mrnm = mtid
        self.qname = "r%d" % index
        self.tags = {}
        self.index = index

    def has_tag(self, key):
        return key in self.tags

    def set_tag(self, key, value, value_type=None):
        self.tags[key] = (value, value_type)


def run_reference_rescale(ref, batch, csv_text, len5p, len3p):
    """Runs the reference's own _rescale_qual_core (routing + _rescale_qual_read) with a stand-in
    pysam.AlignmentFile that iterates duck-typed reads and collects what is written.
    Returns (list of new qual lists per record, MR values (None when absent), log lines)."""
    import logging
    import tempfile
    md = import_reference()
    import pysam as pysam_stub

    reads = [_RescaleRead(batch.record(i), i, int(batch.mtid[i]), int(batch.mpos[i])) for i in range(batch.n)]
    written = []

    class FakeAlignmentFile:
        def __init__(self, path, mode="r", template=None):
            self.mode = mode

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

        def __iter__(self):
            return iter(reads)

        def getrname(self, tid):
            return ref.names[tid]

        def write(self, hit):
            written.append(hit)

    pysam_stub.AlignmentFile = FakeAlignmentFile
    records = []

    class Capture(logging.Handler):
        def emit(self, record):
            records.append(record.getMessage())

    handler = Capture()
    logging.getLogger("mapdamage.rescale").addHandler(handler)
    logging.getLogger("mapdamage.rescale").setLevel(logging.INFO)
    with tempfile.TemporaryDirectory() as tmp:
        folder = pathlib.Path(tmp)
        (folder / "Stats_out_MCMC_correct_prob.csv").write_text(csv_text)
        options = types.SimpleNamespace(folder=folder, rescale_length_5p=len5p, rescale_length_3p=len3p,
                                        filename="in.bam", rescale_out="out.bam")
        md.rescale._rescale_qual_core(_Fasta(ref), options)
    logging.getLogger("mapdamage.rescale").removeHandler(handler)
    assert len(written) == batch.n
    quals, mrs = [], []
    for hit in written:
        quals.append(None if hit.qual is None else [ord(c) - 33 for c in hit.qual])
        mrs.append(hit.tags["MR"][0] if "MR" in hit.tags else None)
    return quals, mrs, records
