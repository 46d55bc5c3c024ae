"""Quality rescaling (mapdamage/rescale.py; BASELINE config[4]).

The reference lowers the base quality of every C>T (5' side) / G>A (3' side) column by the
posterior damage probability of its position (``_rescale_qual_read``, rescale.py:195-282).  The
new quality depends only on (substitution, position key, old quality), so the floating-point
work — ``10 ** x``, ``math.log10``, Python's half-to-even ``round`` — is done **once on the host,
with the reference's own expressions**, into an integer lookup table; the HIP kernel is a pure
byte lookup and therefore bit-exact.  The ``MR`` tag (sum of the correction probabilities of a
read, rescale.py:246,275) is summed on the device in fp64 in the reference's column order.
"""

import csv
import math
import pathlib

import numpy as np

STATUS_UNMAPPED, STATUS_NO_QUAL, STATUS_BOTH, STATUS_FORWARD, STATUS_IMPROPER = range(5)


class RescaleError(RuntimeError):
    pass


def get_corr_prob(filepath, rescale_length_5p, rescale_length_3p):
    """The posterior damage probabilities the rescaling uses, ``{(ref, read, position): probability}``, from the
    ``Stats_out_MCMC_correct_prob.csv`` the reference's R stage writes (header ``"","Position","C.T","G.A"``; what
    mapdamage/rescale.py:23-46 reads): positions 1 .. ``rescale_length_5p`` from the 5' end, -1 .. -``rescale_length_3p``
    from the 3' end."""
    path = pathlib.Path(filepath)
    if not path.is_file():
        raise RescaleError("File does not exist; please re-run mapDamage")
    wanted = range(-int(rescale_length_3p), int(rescale_length_5p) + 1)
    columns = {"C.T": ("C", "T"), "G.A": ("G", "A")}
    table = {}
    with path.open(newline="") as handle:
        rows = csv.reader(handle, strict=True)
        try:
            header = next(rows, [])
            at = {name: header.index(name) for name in ("Position", *columns)}
            for row in rows:
                position = int(row[at["Position"]])
                if position in wanted:
                    for name, (ref, read) in columns.items():
                        table[(ref, read, position)] = float(row[at[name]])
        except csv.Error as error:
            raise RescaleError("Error while reading line %d: %s" % (rows.line_num, error))
    return table


def _phred_pval_to_raw(pval):
    return int(round(-10 * math.log10(abs(pval))))          # rescale.py:13-15 (without the +33)


def _phred_raw_to_pval(q):
    return 10 ** (-(float(q + 33) - float(33)) / 10)        # rescale.py:18-20


class RescaleModel:
    """Lookup tables indexed [substitution 0=C>T,1=G>A][position key index][old quality 0..93].

    Position key index: 0 = no correction for that key; 1..len5p = positions from the 5' end;
    len5p + k = position -k from the 3' end."""

    def __init__(self, corr_prob, rescale_length_5p, rescale_length_3p):
        self.len5p, self.len3p = int(rescale_length_5p), int(rescale_length_3p)
        self.corr_prob = dict(corr_prob)
        self.npos = 1 + self.len5p + self.len3p
        self.lut = np.zeros((2, self.npos, 94), np.uint8)
        self.term = np.zeros((2, self.npos), np.float64)
        for si, (ref, read) in enumerate((("C", "T"), ("G", "A"))):
            for kidx in range(self.npos):
                position = kidx if kidx <= self.len5p else -(kidx - self.len5p)
                corr = corr_prob.get((ref, read, position), 0) if kidx else 0
                pdam = 1 - corr                                          # rescale.py:232-239
                self.term[si, kidx] = 1 - pdam                           # rescale.py:246
                for q in range(94):
                    pseq = 1 - _phred_raw_to_pval(q)                     # rescale.py:240
                    newp = pdam * pseq                                   # rescale.py:241
                    self.lut[si, kidx, q] = _phred_pval_to_raw(1 - newp)  # rescale.py:242

    @classmethod
    def from_csv(cls, path, rescale_length_5p, rescale_length_3p):
        return cls(get_corr_prob(path, rescale_length_5p, rescale_length_3p), rescale_length_5p,
                   rescale_length_3p)


class RescaleSummary:
    """The ``subs`` dictionary of the reference (rescale.py:82-105) rebuilt from the device counters
    (include/mdx.h, mdx_rescale_summary) and its log lines (``_qual_summary_subs`` :146-157,
    ``_print_subs`` :159-192).  The integer entries are exact.  The six ``-pvals`` sums are
    floating-point sums that the reference accumulates read by read; every term is a function of
    (substitution, position key, old quality), so they are summed here from the occurrence counts with
    ``math.fsum`` — log-only values, printed with four decimals."""

    SUBS = ("CT", "TC", "GA", "AG")

    def __init__(self, words, model):
        import math
        words = np.asarray(words, dtype=np.uint64)
        npos = model.npos
        assert words.shape[0] == 756 + 2 * npos * 94
        self.bases = {b: int(words[i]) for i, b in enumerate("ACGT")}
        hist = words[4:756].reshape(4, 2, 94)
        self.before = {s: [int(v) for v in hist[i, 0]] for i, s in enumerate(self.SUBS)}
        self.after = {s: [int(v) for v in hist[i, 1]] for i, s in enumerate(self.SUBS)}
        keyhist = words[756:].reshape(2, npos, 94)
        pseq = [1 - _phred_raw_to_pval(q) for q in range(94)]           # rescale.py:240 / :113
        self.pvals = {}
        for si, s in enumerate(("CT", "GA")):
            terms = []
            for k in range(npos):
                position = k if k <= model.len5p else -(k - model.len5p)
                corr = model.corr_prob.get(("C" if si == 0 else "G", "T" if si == 0 else "A", position), 0) if k else 0
                pdam = 1 - corr                                          # rescale.py:232-239, the exact expression
                for q in range(94):
                    c = int(keyhist[si, k, q])
                    if c:
                        terms.append(c * (pdam * pseq[q]))               # prob_corr = newp (rescale.py:241, :251)
            self.pvals[s + "-pvals"] = math.fsum(terms)
            self.pvals[s + "-pvals_before"] = math.fsum(c * pseq[q] for q, c in enumerate(self.before[s]) if c)
        for s in ("TC", "AG"):
            self.pvals[s + "-pvals"] = math.fsum(c * pseq[q] for q, c in enumerate(self.before[s]) if c)

    def quality_level(self, table, sub, level):
        return sum(table[sub][level:])                                   # rescale.py:146-157

    def log_lines(self):
        lines = ["Expected substition frequencies before and after rescaling:"]
        for sub in self.SUBS:
            base_count = self.bases[sub[0]]
            if base_count:
                pvals = self.pvals[sub + "-pvals"]
                before = self.pvals.get(sub + "-pvals_before", pvals)
                lines.append("    %s>%s    %.4f    %.4f" % (sub[0], sub[1], before / base_count, pvals / base_count))
            else:
                lines.append("\t%s\tNA\t\tNA" % (sub,))
        lines.append("Quality metrics before and after scaling:")
        for sub in ("CT", "GA"):
            for level in (0, 10, 20, 30, 40):
                lines.append("    %s-Q%02i% 10i% 10i" % (sub, level, self.quality_level(self.before, sub, level),
                                                      self.quality_level(self.after, sub, level)))
        return lines


def finalize_mr(raw_sum):
    """``float("%.5f" % sum)`` (rescale.py:276)."""
    return float("%.5f" % raw_sum)


def round_mr(mr_raw):
    """``float("%.5f" % x)`` (rescale.py:275-276) element by element, as the float32 of an ``MR:f`` tag; NaN (a record written
    back unchanged) gives 0.  In the library, on the host's threads (include/mdx.h ``mdx_mr_round``)."""
    import ctypes
    from .engine import load_library
    from .sam import usable_cpus
    mr_raw = np.ascontiguousarray(mr_raw, np.float64)
    out = np.zeros(mr_raw.shape[0], np.float32)
    lib = load_library()
    lib.mdx_mr_round.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32]
    if lib.mdx_mr_round(mr_raw.ctypes.data, mr_raw.shape[0], out.ctypes.data, min(32, usable_cpus())) != 0:
        raise ValueError("mdx_mr_round failed")
    return out


def rescale_bam_on_device(engine, ref, in_path, out_path, model, slab_bytes=256 << 20, timings=None):
    """``rescale_bam`` with the records never on the host (include/mdx.h ``mdx_gbam_rescale_slab``): the compressed file goes to
    HBM a slab at a time, is inflated and unpacked there (the decode pipeline of the tabulation pass, with quality and mate
    columns), the rescale kernels list the quality bytes that change, those go into the inflated records themselves, every
    record — a rescaled one with its ``MR:f`` tag — is laid out as the output stream in HBM and compressed there; only BGZF
    members come back.  Same return value; raises ``sam.GpuDecodeUnsupported`` (or ValueError for a damaged file) for what
    the device decoder does not take: the caller falls back to ``rescale_bam``."""
    import time
    from .sam import BgzfWriter, GpuBamStream, bam_header_bytes
    engine.set_reference(ref)
    engine.set_rescale_model(model)
    counts = np.zeros(5, np.int64)
    spent = {"decode": 0.0, "rescale_write_back_deflate": 0.0, "file_write": 0.0}
    clock = time.perf_counter
    # (no read group is looked up: the rescaling pass knows no libraries, rescale.py:285-365)
    with GpuBamStream(engine, in_path, readgroups=[], lib_default=0, chunk_bytes=slab_bytes, want_qual=True, want_mate=True,
                      packed=False) as stream, BgzfWriter(out_path, engine=engine) as out:
        out.write(bam_header_bytes(stream.header))
        while True:
            t0 = clock()
            view = stream.next_view()
            engine.sync()
            t1 = clock()
            spent["decode"] += t1 - t0
            if view is None:
                break
            members = stream.rescale_slab(view, counts)
            t2 = clock()
            spent["rescale_write_back_deflate"] += t2 - t1
            out.write_members(members)
            spent["file_write"] += clock() - t2
        t3 = clock()
    spent["file_write"] += clock() - t3
    if timings is not None:
        timings.update(spent)
    summary = RescaleSummary(engine.rescale_summary(), model)
    return summary, {name: int(counts[code]) for name, code in
            (("unmapped", STATUS_UNMAPPED), ("without_qualities", STATUS_NO_QUAL), ("single_end", STATUS_BOTH),
             ("inward_pairs", STATUS_FORWARD), ("improper_pairs", STATUS_IMPROPER))}


def rescale_bam(engine, ref, in_path, out_path, model, chunk_bytes=256 << 20, timings=None, device_deflate=True):
    """File-level mirror of ``_rescale_qual_core`` (rescale.py:285-365): every record of the BAM is
    written back, rescaled records get their new qualities and an ``MR:f`` tag, everything else in
    the record (name, MAPQ, mate fields, other tags) is preserved byte for byte.
    The file goes through in chunks: native decode (the encoded records kept), one rescale launch per chunk, the
    records patched natively, BGZF blocks deflated on a thread pool — host memory is bounded by the chunk and no
    per-record Python work is done.  Returns (substitution summary, per-status record counts).  ``timings`` (a dict): receives
    the seconds spent waiting for the decoder (BGZF inflate + unpack on the host's threads — the next chunk is decoded on a
    helper thread under this one's work), in the rescaling call (columns to HBM, the kernels, qualities / MR / status back),
    formatting MR, patching the records and deflating + writing the output.  ``device_deflate``: the output's BGZF members are
    made on the device (``mdx_bgzf_deflate``; False: zlib level 6 on the host's threads, as htslib behind the reference does
    — the output's records are the same either way, its compressed bytes are not)."""
    import time
    from .sam import BamStream, BgzfWriter, bam_header_bytes
    engine.set_reference(ref)
    engine.set_rescale_model(model)
    counts = np.zeros(5, np.int64)
    spent = {"decode": 0.0, "rescale": 0.0, "mr_format": 0.0, "patch": 0.0, "deflate_write": 0.0}
    clock = time.perf_counter
    from concurrent.futures import ThreadPoolExecutor
    with BamStream(in_path, chunk_bytes=chunk_bytes, keep_raw=True) as stream, \
            BgzfWriter(out_path, engine=engine if device_deflate else None) as out, ThreadPoolExecutor(1) as ahead:
        out.write(bam_header_bytes(stream.header))
        coming = ahead.submit(stream.next_chunk)
        while True:
            t0 = clock()
            chunk = coming.result()
            if chunk is not None:
                coming = ahead.submit(stream.next_chunk)        # (decoded while this chunk is rescaled, patched and written)
            t1 = clock()
            spent["decode"] += t1 - t0
            if chunk is None:
                break
            batch = chunk.batch
            qual_out, mr_raw, status = engine.rescale(batch)
            t2 = clock()
            spent["rescale"] += t2 - t1
            rescaled = (status == STATUS_BOTH) | (status == STATUS_FORWARD)
            clash = rescaled & (np.asarray(chunk.has_mr) != 0)
            if clash.any():      # rescale.py:277-278
                raise SystemExit("Read: %s already has a MR tag, can't rescale" % chunk.qname_at(int(np.nonzero(clash)[0][0])))
            mr = round_mr(mr_raw)
            t3 = clock()
            spent["mr_format"] += t3 - t2
            patched = stream.patch_rescaled(chunk, qual_out, mr, rescaled)
            t4 = clock()
            spent["patch"] += t4 - t3
            out.write(patched)
            spent["deflate_write"] += clock() - t4
            counts += np.bincount(status, minlength=5)[:5]
        t5 = clock()
    spent["deflate_write"] += clock() - t5          # (the writer's last blocks and its close)
    if timings is not None:
        timings.update(spent)
    summary = RescaleSummary(engine.rescale_summary(), model)
    return summary, {name: int(counts[code]) for name, code in
            (("unmapped", STATUS_UNMAPPED), ("without_qualities", STATUS_NO_QUAL), ("single_end", STATUS_BOTH),
             ("inward_pairs", STATUS_FORWARD), ("improper_pairs", STATUS_IMPROPER))}
