"""Host-side mirror of mapdamage/reader.py: read-group -> (sample, library) mapping, the flag
filter and the two downsampling modes (which consume the Python RNG and therefore stay on
the host)."""

import logging
import random

import numpy as np

from . import layout as L
from .sam import Alignments, BAMError, read_alignments


class BAMReader:
    def __init__(self, filepath, merge_libraries=False, downsample_to=None, downsample_seed=None):
        log = logging.getLogger(__name__)
        self.filepath = filepath
        self.downsample_to = downsample_to
        self.downsample_seed = downsample_seed
        self.is_stream = str(filepath) == "-"
        self.handle: Alignments = read_alignments(filepath)
        self._merge_libraries = merge_libraries
        self._readgroups = {}
        self._libraries = {}
        if merge_libraries:
            self._readgroups[None] = ("*", "*")
            self._libraries[("*", "*")] = {None}
        else:
            self._readgroups = self._collect_readgroups(log, self.handle)
            for readgroup, library in self._readgroups.items():
                self._libraries.setdefault(library, set()).add(readgroup)
        log.info("Found %i libraries in BAM file", len(self._libraries))

    def get_references(self):
        return dict(zip(self.handle.header.references, self.handle.header.lengths))

    def get_libraries(self):
        return list(self._libraries.keys())

    @classmethod
    def _collect_readgroups(cls, log, handle):
        readgroups = {}
        for readgroup in handle.header.get("RG", ()):
            try:
                readgroups[readgroup["ID"]] = (readgroup["SM"], readgroup["LB"])
            except KeyError as error:
                raise BAMError("Incomplete readgroup found: %s is missing %s. Either fix BAM or use "
                               "--merge-libraries" % (readgroup.get("ID", "Unnamed readgroup"), error))
        return readgroups

    def kept_indices(self):
        """Indices of the records the reference would iterate over, in its order
        (reader.py:83-96, 121-164)."""
        flag = self.handle.batch.flag
        kept = np.nonzero((flag & L.FLAG_FILTER) == 0)[0]
        if self.downsample_to is None:
            return kept
        rand = random.Random(self.downsample_seed)
        if self.downsample_to < 1:
            return np.asarray([i for i in kept if rand.random() < self.downsample_to], dtype=np.int64)
        size = int(self.downsample_to)
        sample = [None] * size
        for index, record in enumerate(kept):
            if index >= size:
                index = rand.randint(0, index)
                if index >= size:
                    continue
            sample[index] = record
        result = [r for r in sample if r is not None]
        b = self.handle.batch
        result.sort(key=lambda r: (int(b.tid[r]), int(b.pos[r])))
        return np.asarray(result, dtype=np.int64)

    def library_column(self, indices):
        """Library id (index into ``get_libraries()``) of each selected record; raises
        ``BAMError`` like reader.py:63-81 for a missing or unknown read group."""
        libs = self.get_libraries()
        if self._merge_libraries:
            return np.zeros(len(indices), np.uint16)
        index_of = {rg: libs.index(lib) for rg, lib in self._readgroups.items()}
        h = self.handle
        if h._rg is None and h.rg_index is not None:
            # native decoder: read groups are small integers already — one table lookup for all records
            ri = h.rg_index[np.asarray(indices, dtype=np.int64)]
            lut = np.asarray([index_of.get(name, -1) for name in h.rg_names] + [-1], dtype=np.int64)
            lib = lut[ri]     # (index -1 = no RG tag -> the trailing -1)
            bad = np.nonzero(lib < 0)[0]
            if bad.size:
                self._raise_readgroup(int(indices[int(bad[0])]), None if ri[bad[0]] < 0 else h.rg_names[int(ri[bad[0]])])
            return lib.astype(np.uint16)
        out = np.zeros(len(indices), np.uint16)
        for k, i in enumerate(indices):
            rg = h.rg[i]
            if rg is None or rg not in index_of:
                self._raise_readgroup(i, rg)
            out[k] = index_of[rg]
        return out

    def _raise_readgroup(self, i, rg):
        if rg is None:
            raise BAMError("Read %r has no read-group. Either fix BAM or use --merge-libraries"
                           % (self.handle.qname_at(i),))
        raise BAMError("Read %r has read-group not listed in BAM header (%r); either fix BAM "
                       "or use --merge-libraries" % (self.handle.qname_at(i), rg))
