"""SoA column buffers for a batch of alignment records (the host side of the boundary).

One ``ReadBatch`` holds exactly what the reference's loop body reads from each
``pysam.AlignedSegment`` (mapdamage/main.py:165-217): flag, tid, pos,
template length, library id, CIGAR ops in BAM encoding (``len << 4 | op``), the *full*
SEQ bytes (soft clips included, as a BAM record stores them; the aligned query of
pysam's ``read.query`` is derived from the CIGAR on the device) and, optionally, raw
Phred qualities (BAM convention, 0xFF in the first byte = no qualities).
"""

from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import layout as L


@dataclass
class ReadBatch:
    flag: np.ndarray        # u16[n]
    lib: np.ndarray         # u16[n]
    tid: np.ndarray         # i32[n]
    pos: np.ndarray         # i32[n]
    tlen: np.ndarray        # i32[n]
    cigar_off: np.ndarray   # u32[n+1]
    cigar: np.ndarray       # u32[n_ops]  (len << 4 | op)
    seq_off: np.ndarray     # u32[n+1]
    seq: np.ndarray         # u8[n_bases] ASCII
    qual: Optional[np.ndarray] = None  # u8[n_bases] raw phred, or None
    mtid: Optional[np.ndarray] = None  # i32[n] mate reference id (only the rescale routing reads it)
    mpos: Optional[np.ndarray] = None  # i32[n] mate position

    @property
    def n(self):
        return int(self.flag.shape[0])

    def validate(self):
        n = self.n
        assert self.flag.dtype == np.uint16 and self.lib.dtype == np.uint16
        assert self.tid.dtype == np.int32 and self.pos.dtype == np.int32
        assert self.tlen.dtype == np.int32
        assert self.cigar_off.dtype == np.uint32 and self.cigar_off.shape == (n + 1,)
        assert self.seq_off.dtype == np.uint32 and self.seq_off.shape == (n + 1,)
        assert self.cigar.dtype == np.uint32 and self.seq.dtype == np.uint8
        assert int(self.cigar_off[-1]) == self.cigar.shape[0] if n else True
        assert int(self.seq_off[-1]) == self.seq.shape[0] if n else True
        if self.qual is not None:
            assert self.qual.dtype == np.uint8 and self.qual.shape == self.seq.shape
        return self

    def slice(self, lo, hi, copy=True):
        """Records [lo, hi) as a batch of their own (offsets rebased).  ``copy=False``: the columns are views of
        this batch (only the two rebased offset columns are new arrays)."""
        lo = max(0, lo)
        hi = min(self.n, hi)
        if hi < lo:
            hi = lo
        c0, c1 = int(self.cigar_off[lo]), int(self.cigar_off[hi])
        s0, s1 = int(self.seq_off[lo]), int(self.seq_off[hi])
        own = (lambda a: a.copy()) if copy else (lambda a: a)
        return ReadBatch(
            flag=own(self.flag[lo:hi]),
            lib=own(self.lib[lo:hi]),
            tid=own(self.tid[lo:hi]),
            pos=own(self.pos[lo:hi]),
            tlen=own(self.tlen[lo:hi]),
            cigar_off=(self.cigar_off[lo:hi + 1] - np.uint32(c0)).astype(np.uint32),
            cigar=own(self.cigar[c0:c1]),
            seq_off=(self.seq_off[lo:hi + 1] - np.uint32(s0)).astype(np.uint32),
            seq=own(self.seq[s0:s1]),
            qual=None if self.qual is None else own(self.qual[s0:s1]),
            mtid=None if self.mtid is None else own(self.mtid[lo:hi]),
            mpos=None if self.mpos is None else own(self.mpos[lo:hi]),
        )

    def take(self, index):
        """Records selected by an integer index array, in that order (vectorised gather of the ragged columns)."""
        index = np.asarray(index, dtype=np.int64)

        def ragged(off, data):
            off = off.astype(np.int64)
            lens = off[index + 1] - off[index]
            new_off = np.zeros(index.shape[0] + 1, np.int64)
            np.cumsum(lens, out=new_off[1:])
            # element k of record j comes from off[index[j]] + k
            src = np.arange(int(new_off[-1]), dtype=np.int64) + np.repeat(off[index] - new_off[:-1], lens)
            return new_off.astype(np.uint32), [d[src] if d is not None else None for d in data]

        cigar_off, (cigar,) = ragged(self.cigar_off, [self.cigar])
        seq_off, (seq, qual) = ragged(self.seq_off, [self.seq, self.qual])
        return ReadBatch(flag=self.flag[index], lib=self.lib[index], tid=self.tid[index], pos=self.pos[index],
                         tlen=self.tlen[index], cigar_off=cigar_off, cigar=cigar, seq_off=seq_off, seq=seq, qual=qual,
                         mtid=None if self.mtid is None else self.mtid[index],
                         mpos=None if self.mpos is None else self.mpos[index])

    def shard(self, rank, world):
        """Contiguous shard ``rank`` of ``world`` (shard-by-read)."""
        n = self.n
        lo = (n * rank) // world
        hi = (n * (rank + 1)) // world
        return self.slice(lo, hi)

    def record(self, i):
        """Record ``i`` as a plain dict (used by the golden generator and the tests)."""
        c0, c1 = int(self.cigar_off[i]), int(self.cigar_off[i + 1])
        s0, s1 = int(self.seq_off[i]), int(self.seq_off[i + 1])
        ops = [(int(c) & 0xF, int(c) >> 4) for c in self.cigar[c0:c1]]
        rec = dict(
            flag=int(self.flag[i]),
            lib=int(self.lib[i]),
            tid=int(self.tid[i]),
            pos=int(self.pos[i]),
            tlen=int(self.tlen[i]),
            cigar=ops,
            seq=self.seq[s0:s1].tobytes().decode("latin-1"),
            qual=None,
        )
        if self.qual is not None and s1 > s0 and int(self.qual[s0]) != L.QUAL_MISSING:
            rec["qual"] = [int(q) for q in self.qual[s0:s1]]
        return rec

    def nbytes(self):
        tot = 0
        for a in (self.flag, self.lib, self.tid, self.pos, self.tlen, self.cigar_off,
                  self.cigar, self.seq_off, self.seq, self.qual):
            if a is not None:
                tot += a.nbytes
        return tot


def batch_from_records(records, with_qual=None):
    """Pack a list of record dicts (see ``ReadBatch.record``) into SoA columns."""
    n = len(records)
    if with_qual is None:
        with_qual = any(r.get("qual") is not None for r in records)
    flag = np.zeros(n, np.uint16)
    lib = np.zeros(n, np.uint16)
    tid = np.zeros(n, np.int32)
    pos = np.zeros(n, np.int32)
    tlen = np.zeros(n, np.int32)
    cigar_off = np.zeros(n + 1, np.uint32)
    seq_off = np.zeros(n + 1, np.uint32)
    cig, seqs, quals = [], [], []
    co = so = 0
    for i, r in enumerate(records):
        flag[i] = r["flag"]
        lib[i] = r.get("lib", 0)
        tid[i] = r["tid"]
        pos[i] = r["pos"]
        tlen[i] = r.get("tlen", 0)
        for op, ln in r["cigar"]:
            cig.append((ln << 4) | op)
        co += len(r["cigar"])
        cigar_off[i + 1] = co
        s = r["seq"].encode("latin-1")
        seqs.append(s)
        if with_qual:
            q = r.get("qual")
            if q is None:
                quals.append(bytes([L.QUAL_MISSING]) * len(s))
            else:
                assert len(q) == len(s)
                quals.append(bytes(q))
        so += len(s)
        seq_off[i + 1] = so
    seq = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    qual = np.frombuffer(b"".join(quals), dtype=np.uint8).copy() if with_qual else None
    return ReadBatch(flag, lib, tid, pos, tlen, cigar_off,
                     np.asarray(cig, dtype=np.uint32), seq_off, seq, qual).validate()


FLAG_QUAL_ABOVE_MIN = 0x8000     # include/mdx.h MDX_FLAG_QUAL_ABOVE_MIN


def record_min_quality(batch):
    """Lowest quality of each record (0xFF for a record without bases or without qualities)."""
    if batch.qual is None:
        return np.full(batch.n, 0xFF, np.uint8)
    out = np.full(batch.n, 0xFF, np.uint8)
    off = batch.seq_off.astype(np.int64)
    nonempty = off[1:] > off[:-1]
    if nonempty.any():
        out[nonempty] = np.minimum.reduceat(batch.qual[:int(off[-1])], off[:-1][nonempty])
    return out


def mark_unmaskable(batch, minqual, qmin=None):
    """Set the MDX_FLAG_QUAL_ABOVE_MIN hint on the records none of whose qualities is below ``minqual`` (their
    quality windows are not loaded by the kernel) and say whether *no* record of the batch can be masked — the batch
    may then be tabulated without its quality column (the unmasked kernel).  Returns (batch, nothing_to_mask)."""
    if not minqual or batch.qual is None:
        return batch, True
    qmin = record_min_quality(batch) if qmin is None else np.asarray(qmin)
    fine = qmin >= minqual            # (0xFF: no qualities at all — never masked either)
    # the bit is this function's to set: whatever the column held there before (a FLAG with bit 15 set in a file) is not a hint
    base = batch.flag & np.uint16(0x7FFF)
    batch.flag = np.where(fine, base | np.uint16(FLAG_QUAL_ABOVE_MIN), base).astype(np.uint16)
    return batch, bool(fine.all())


def concat_batches(batches):
    batches = [b for b in batches if b.n]
    if not batches:
        return batch_from_records([])
    with_qual = all(b.qual is not None for b in batches)
    coffs, soffs = [np.zeros(1, np.uint32)], [np.zeros(1, np.uint32)]
    cbase = sbase = 0
    for b in batches:
        coffs.append((b.cigar_off[1:].astype(np.int64) + cbase).astype(np.uint32))
        soffs.append((b.seq_off[1:].astype(np.int64) + sbase).astype(np.uint32))
        cbase += int(b.cigar_off[-1])
        sbase += int(b.seq_off[-1])
    cat = np.concatenate
    return ReadBatch(
        cat([b.flag for b in batches]), cat([b.lib for b in batches]),
        cat([b.tid for b in batches]), cat([b.pos for b in batches]),
        cat([b.tlen for b in batches]), cat(coffs), cat([b.cigar for b in batches]),
        cat(soffs), cat([b.seq for b in batches]),
        cat([b.qual for b in batches]) if with_qual else None,
    ).validate()


@dataclass
class Reference:
    """Contigs in BAM ``tid`` order; bases kept as the original FASTA bytes."""
    names: list
    seqs: list  # list of bytes (original case)

    @property
    def lengths(self):
        return [len(s) for s in self.seqs]

    def concat(self):
        """(bases u8[total], contig_off i64[n+1]); built once per object (a genome of human size is not joined again
        for every caller).  Both arrays are shared between callers and read-only: they must not be modified."""
        cat = self.__dict__.get("_cat")
        # (the cache holds the contig objects it was built from: none of them can have been freed and its address
        # reused; a replaced contig is another object)
        if cat is None or len(cat[2]) != len(self.seqs) or any(a is not b for a, b in zip(cat[2], self.seqs)):
            offs = np.zeros(len(self.seqs) + 1, np.int64)
            for i, s in enumerate(self.seqs):
                offs[i + 1] = offs[i] + len(s)
            bases = np.frombuffer(b"".join(self.seqs), dtype=np.uint8)
            offs.setflags(write=False)
            cat = self.__dict__["_cat"] = (bases, offs, list(self.seqs))
        return cat[0], cat[1]
