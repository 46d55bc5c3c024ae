"""Host-side mirror of mapdamage/reader.py: read-group -> (sample, library) mapping, the flag
filter and the two downsampling modes (which consume the Python RNG and therefore stay on
the host)."""

import logging
import random

import numpy as np

from . import layout as L
from .batch import ReadBatch
from .sam import Alignments, BamStream, BAMError, is_bam, read_alignments


def draw_uniform(rand, n):
    """The next ``n`` values of ``rand.random()`` (a ``random.Random``), drawn at once: numpy's legacy generator is the same
    Mersenne Twister with the same 53-bit conversion, so the state goes over, ``n`` doubles come back and the state
    returns — the stream of draws of reader.py:139-141 without a Python call per record."""
    st = rand.getstate()
    rs = np.random.RandomState()
    rs.set_state(("MT19937", np.array(st[1][:-1], dtype=np.uint32), st[1][-1]))
    out = rs.random_sample(int(n))
    ns = rs.get_state()
    rand.setstate((st[0], tuple(int(x) for x in ns[1]) + (int(ns[2]),), st[2]))
    return out


def downsample_indices(flag, tid, pos, downsample_to, rand):
    """Indices of the records of one stretch of a file that the reference iterates over (reader.py:83-96): the flag filter
    (reader.py:121-132), then — ``downsample_to`` below 1 — one draw of ``rand`` per kept record in file order
    (reader.py:134-146), or — 1 and more — reservoir sampling and a stable sort by (tid, pos) (reader.py:148-164).
    ``rand``: the ``random.Random`` that carries the draws from stretch to stretch."""
    kept = np.nonzero((np.asarray(flag) & L.FLAG_FILTER) == 0)[0]
    if downsample_to is None:
        return kept
    if downsample_to < 1:
        return kept[draw_uniform(rand, len(kept)) < downsample_to].astype(np.int64)
    size = int(downsample_to)
    sample = [None] * size
    for index, record in enumerate(kept):
        if index >= size:
            index = rand.randint(0, index)
            if index >= size:
                continue
        sample[index] = record
    result = [r for r in sample if r is not None]
    result.sort(key=lambda r: (int(tid[r]), int(pos[r])))
    return np.asarray(result, dtype=np.int64)


class BAMReader:
    def __init__(self, filepath, merge_libraries=False, downsample_to=None, downsample_seed=None,
                 chunk_bytes=None):
        """``chunk_bytes``: decode a BAM file in chunks of that many uncompressed bytes (``iter_batches``
        then yields one batch per chunk and ``handle`` holds the header only) instead of all at once.
        Downsampling to a fixed number of reads needs the whole file (reservoir + coordinate sort,
        reader.py:148-164) and SAM text is small-file territory: both keep the one-piece decode."""
        log = logging.getLogger(__name__)
        self.filepath = filepath
        self.downsample_to = downsample_to
        self.downsample_seed = downsample_seed
        self.is_stream = str(filepath) == "-"
        self._chunks = None
        if chunk_bytes and is_bam(filepath) and (downsample_to is None or downsample_to < 1):
            self._chunks = BamStream(filepath, chunk_bytes=chunk_bytes)
            empty = ReadBatch(np.zeros(0, np.uint16), np.zeros(0, np.uint16), np.zeros(0, np.int32),
                              np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(1, np.uint32),
                              np.zeros(0, np.uint32), np.zeros(1, np.uint32), np.zeros(0, np.uint8),
                              np.zeros(0, np.uint8))
            self.handle = Alignments(self._chunks.header, empty, [], [])
        else:
            self.handle: Alignments = read_alignments(filepath)
        self._merge_libraries = merge_libraries
        # read group id -> (sample, library); with --merge-libraries every record, tagged or not, is ("*", "*")
        self._readgroups = {None: ("*", "*")} if merge_libraries else self._collect_readgroups(log, self.handle)
        # libraries in first-appearance order (the order of the header's @RG lines), each with its read groups
        self._libraries = {}
        for readgroup, library in self._readgroups.items():
            self._libraries.setdefault(library, set()).add(readgroup)
        log.info("Found %i libraries in BAM file", len(self._libraries))

    def get_references(self):
        return dict(zip(self.handle.header.references, self.handle.header.lengths))

    def get_libraries(self):
        return list(self._libraries.keys())

    @staticmethod
    def _collect_readgroups(log, handle):
        """{ID: (SM, LB)} of the header's @RG lines; a line without one of the three tags is the reference's
        BAMError (reader.py:99-118)."""
        readgroups = {}
        for line in handle.header.get("RG", ()):
            missing = [tag for tag in ("ID", "SM", "LB") if tag not in line]
            if missing:
                raise BAMError("Incomplete readgroup found: %s is missing %s. Either fix BAM or use "
                               "--merge-libraries" % (line.get("ID", "Unnamed readgroup"), KeyError(missing[0])))
            readgroups[line["ID"]] = (line["SM"], line["LB"])
        return readgroups

    def iter_batches(self, resume=None):
        """The records the reference would iterate over (reader.py:83-96, 121-164), in its order, as
        ``ReadBatch``es with the library column filled in: one per decoded chunk, or the whole file
        at once.  The decode of chunk k+1 runs on a helper thread while the caller works on chunk k.
        ``resume``: (compressed offset of a BGZF block, inflated bytes in front of the first record wanted) — the records
        from there on only (chunked BAM decode: ``BamStream.seek``)."""
        if resume is not None:
            if self._chunks is None:
                raise ValueError("resuming needs the chunked BAM decoder")
            self._chunks.seek(*resume)
        if self._chunks is None:
            indices = self.kept_indices()
            batch = self.handle.batch
            if len(indices) != batch.n:
                batch = batch.take(indices)
            batch.lib = self.library_column(indices)
            yield batch
            return
        from concurrent.futures import ThreadPoolExecutor
        rand = random.Random(self.downsample_seed)
        with ThreadPoolExecutor(1) as pool:
            pending = pool.submit(self._chunks.next_chunk)
            while True:
                chunk = pending.result()
                if chunk is None:
                    break
                pending = pool.submit(self._chunks.next_chunk)
                indices = self.kept_indices(chunk, rand)
                batch = chunk.batch
                if len(indices) != batch.n:
                    batch = batch.take(indices)
                batch.lib = self.library_column(indices, chunk)
                yield batch
        self._chunks.close()

    def kept_indices(self, handle=None, rand=None):
        """Indices of the records the reference would iterate over, in its order
        (reader.py:83-96, 121-164).  ``handle``/``rand``: one chunk of a file decoded in pieces, and
        the generator that carries the --downsample stream of draws across chunks."""
        b = (handle or self.handle).batch
        if rand is None:
            rand = random.Random(self.downsample_seed)
        return downsample_indices(b.flag, b.tid, b.pos, self.downsample_to, rand)

    def library_column(self, indices, handle=None):
        """Library id (index into ``get_libraries()``) of each selected record; raises
        ``BAMError`` like reader.py:63-81 for a missing or unknown read group."""
        libs = self.get_libraries()
        if self._merge_libraries:
            return np.zeros(len(indices), np.uint16)
        index_of = {rg: libs.index(lib) for rg, lib in self._readgroups.items()}
        h = handle or self.handle
        if h._rg is None and h.rg_index is not None:
            # native decoder: read groups are small integers already — one table lookup for all records
            ri = h.rg_index[np.asarray(indices, dtype=np.int64)]
            lut = np.asarray([index_of.get(name, -1) for name in h.rg_names] + [-1], dtype=np.int64)
            lib = lut[ri]     # (index -1 = no RG tag -> the trailing -1)
            bad = np.nonzero(lib < 0)[0]
            if bad.size:
                self._raise_readgroup(int(indices[int(bad[0])]), None if ri[bad[0]] < 0 else h.rg_names[int(ri[bad[0]])], h)
            return lib.astype(np.uint16)
        out = np.zeros(len(indices), np.uint16)
        for k, i in enumerate(indices):
            rg = h.rg[i]
            if rg is None or rg not in index_of:
                self._raise_readgroup(i, rg, h)
            out[k] = index_of[rg]
        return out

    def _raise_readgroup(self, i, rg, handle=None):
        handle = handle or self.handle
        if rg is None:
            raise BAMError("Read %r has no read-group. Either fix BAM or use --merge-libraries"
                           % (handle.qname_at(i),))
        raise BAMError("Read %r has read-group not listed in BAM header (%r); either fix BAM "
                       "or use --merge-libraries" % (handle.qname_at(i), rg))
