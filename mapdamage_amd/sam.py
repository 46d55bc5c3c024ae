"""SAM / BAM decoding and encoding without pysam (pysam is absent from this image).

Produces the SoA ``ReadBatch`` columns of the boundary directly.  Text SAM and BGZF-compressed
BAM (through the standard library's zlib) are supported for reading and writing; this is the
host-side "N1" row of SURVEY §8f in its first, pure-Python form.
"""

import gzip
import io
import struct
import zlib

import numpy as np

from . import layout as L
from .batch import ReadBatch

_SEQ_DECODE = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
_SEQ_ENCODE = np.full(256, 15, np.uint8)
for _i, _c in enumerate(b"=ACMGRSVTWYHKDBN"):
    _SEQ_ENCODE[_c] = _i
    _SEQ_ENCODE[ord(chr(_c).lower())] = _i


def usable_cpus():
    """CPUs this process may keep busy: its affinity mask, cut down to what the control group grants (``cpu.max``: a pod
    with 256 hardware threads and 16 CPUs' worth of quota stalls for most of every period under 64 busy threads;
    ``MDX_CPU_MAX_FILE`` names a stand-in for the tests) and divided by the ranks of this node (``LOCAL_WORLD_SIZE``, as
    torchrun sets it: one process per GPU, SURVEY 8e, and all of them decode at the same time).  The same rule as the
    library's own pool (include/mdx.h ``mdx_host_threads``)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open(os.environ.get("MDX_CPU_MAX_FILE") or "/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max" and int(period) > 0:
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    try:
        ranks = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    except ValueError:
        ranks = 1
    if ranks > 1:
        n //= ranks
    return max(1, n)


class BAMError(RuntimeError):
    """Read-group problems (mapdamage/reader.py:16-17)."""


class Header:
    def __init__(self, text=""):
        self.text = text
        self.references, self.lengths, self.read_groups = [], [], []
        for line in text.splitlines():
            fields = line.split("\t")
            tags = dict(f.split(":", 1) for f in fields[1:] if ":" in f)
            if fields[0] == "@SQ":
                self.references.append(tags["SN"])
                self.lengths.append(int(tags["LN"]))
            elif fields[0] == "@RG":
                self.read_groups.append(tags)

    def get(self, key, default=()):
        return self.read_groups if key == "RG" else default


class Alignments:
    """All records of a SAM/BAM file: header + SoA batch + per-record RG id and query name.
    ``raw`` (BAM only, ``keep_raw=True``): the record bodies as stored, and ``raw_header`` the bytes
    in front of the first record, so that a rewritten file keeps every field it does not touch."""

    def __init__(self, header, batch, rg, qname):
        self.header, self.batch, self._rg, self._qname = header, batch, rg, qname
        self.raw, self.raw_header, self.has_mr = None, None, None
        # native decoder: per-record read-group index (-1 = no RG tag) into rg_names, and the query names as one
        # blob + offsets; the Python lists are only built when somebody asks for them
        self.rg_index, self.rg_names, self._qblob, self._qoff = None, None, None, None

    @property
    def rg(self):
        if self._rg is None:
            names = self.rg_names
            self._rg = [names[i] if i >= 0 else None for i in self.rg_index.tolist()]
        return self._rg

    @rg.setter
    def rg(self, value):
        self._rg = value

    @property
    def qname(self):
        if self._qname is None:
            blob, off = bytes(self._qblob), self._qoff.tolist()
            self._qname = [blob[off[i]:off[i + 1]].decode() for i in range(len(off) - 1)]
        return self._qname

    @qname.setter
    def qname(self, value):
        self._qname = value

    def qname_at(self, i):
        if self._qname is not None:
            return self._qname[i]
        return bytes(self._qblob[int(self._qoff[i]):int(self._qoff[i + 1])]).decode()


def _finish(header, flags, tids, poss, tlens, cigs, cig_counts, seqs, quals, rgs, qnames):
    n = len(flags)
    cigar_off = np.zeros(n + 1, np.uint32)
    np.cumsum(np.asarray(cig_counts, dtype=np.int64), out=cigar_off[1:])
    seq_off = np.zeros(n + 1, np.uint32)
    np.cumsum(np.asarray([len(s) for s in seqs], dtype=np.int64), out=seq_off[1:])
    seq = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    qual = np.frombuffer(b"".join(quals), dtype=np.uint8).copy()
    batch = ReadBatch(np.asarray(flags, np.uint16), np.zeros(n, np.uint16), np.asarray(tids, np.int32),
                      np.asarray(poss, np.int32), np.asarray(tlens, np.int32), cigar_off,
                      np.asarray(cigs, np.uint32), seq_off, seq, qual).validate()
    return Alignments(header, batch, rgs, qnames)


def read_sam(path_or_handle):
    handle = open(path_or_handle, "rt") if isinstance(path_or_handle, (str, bytes)) or hasattr(path_or_handle, "__fspath__") else path_or_handle
    head, lines = [], []
    for line in handle:
        (head if line.startswith("@") else lines).append(line)
    header = Header("".join(head))
    tid_of = {name: i for i, name in enumerate(header.references)}
    flags, tids, poss, tlens, cigs, cig_counts, seqs, quals, rgs, qnames = [], [], [], [], [], [], [], [], [], []
    for line in lines:
        f = line.rstrip("\n").split("\t")
        if len(f) < 11:
            continue
        qnames.append(f[0])
        flags.append(int(f[1]) & 0x3FFF)       # bits 14 and 15 are the kernels' hint bits (mdx.h MDX_FLAG_HAS_QUAL / _QUAL_ABOVE_MIN), never the file's
        tids.append(tid_of.get(f[2], -1))
        poss.append(int(f[3]) - 1)
        tlens.append(int(f[8]))
        n_ops, num = 0, 0
        if f[5] != "*":
            for ch in f[5]:
                if ch.isdigit():
                    num = num * 10 + ord(ch) - 48
                else:
                    cigs.append((num << 4) | L.CIGAR_CHARS.index(ch))
                    n_ops += 1
                    num = 0
        cig_counts.append(n_ops)
        s = b"" if f[9] == "*" else f[9].upper().encode()
        # htslib stores SEQ through the 4-bit alphabet: anything else becomes N
        s = bytes(_SEQ_DECODE[_SEQ_ENCODE[np.frombuffer(s, dtype=np.uint8)]])
        seqs.append(s)
        if f[10] == "*":
            quals.append(b"\xff" * len(s))
        else:
            quals.append(bytes((np.frombuffer(f[10].encode(), dtype=np.uint8) - 33).astype(np.uint8)))
        rg = None
        for tag in f[11:]:
            if tag.startswith("RG:Z:"):
                rg = tag[5:]
        rgs.append(rg)
    return _finish(header, flags, tids, poss, tlens, cigs, cig_counts, seqs, quals, rgs, qnames)


def read_bam(path, keep_raw=False):
    with gzip.open(path, "rb") as handle:   # BGZF is a series of gzip members
        data = handle.read()
    if data[:4] != b"BAM\x01":
        raise ValueError("%r is not a BAM file" % (path,))
    l_text, = struct.unpack_from("<i", data, 4)
    text = data[8:8 + l_text].split(b"\x00")[0].decode()
    off = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, off)
    off += 4
    names, lengths = [], []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, off)
        names.append(data[off + 4:off + 4 + l_name - 1].decode())
        l_ref, = struct.unpack_from("<i", data, off + 4 + l_name)
        lengths.append(l_ref)
        off += 8 + l_name
    header = Header(text)
    if not header.references:
        header.references, header.lengths = names, lengths
    flags, tids, poss, tlens, cigs, cig_counts, seqs, quals, rgs, qnames = [], [], [], [], [], [], [], [], [], []
    mtids, mposs, raws, has_mr = [], [], [], []
    raw_header = data[:off]
    n = len(data)
    while off + 4 <= n:
        block_size, = struct.unpack_from("<i", data, off)
        rec = memoryview(data)[off + 4:off + 4 + block_size]
        off += 4 + block_size
        tid, pos, l_read_name, _mapq, _bin, n_cigar, flag, l_seq, ntid, npos, tlen = struct.unpack_from("<iiBBHHHiiii", rec, 0)
        mtids.append(ntid)
        mposs.append(npos)
        if keep_raw:
            raws.append(bytes(rec))
        p = 32
        qnames.append(bytes(rec[p:p + l_read_name - 1]).decode())
        p += l_read_name
        cig = np.frombuffer(rec[p:p + 4 * n_cigar], dtype="<u4")
        p += 4 * n_cigar
        packed = np.frombuffer(rec[p:p + (l_seq + 1) // 2], dtype=np.uint8)
        p += (l_seq + 1) // 2
        nib = np.empty(packed.shape[0] * 2, np.uint8)
        nib[0::2] = packed >> 4
        nib[1::2] = packed & 15
        seqs.append(bytes(_SEQ_DECODE[nib[:l_seq]]))
        quals.append(bytes(rec[p:p + l_seq]))
        p += l_seq
        rg = None
        aux = bytes(rec[p:])
        has_mr.append(False)
        q = 0
        while q + 3 <= len(aux):
            tag, typ = aux[q:q + 2], aux[q + 2:q + 3]
            q += 3
            if tag == b"MR":
                has_mr[-1:] = [True]
            if typ == b"Z" or typ == b"H":
                end = aux.index(b"\x00", q)
                if tag == b"RG":
                    rg = aux[q:end].decode()
                q = end + 1
            elif typ in b"AcC":
                q += 1
            elif typ in b"sS":
                q += 2
            elif typ in b"iIf":
                q += 4
            elif typ == b"B":
                sub = aux[q:q + 1]
                cnt, = struct.unpack_from("<i", aux, q + 1)
                if (tag == b"CG" and sub == b"I" and n_cigar == 2 and int(cig[0]) == ((l_seq << 4) | 4) and (int(cig[1]) & 15) == 3):
                    # SAMv1 4.2.2: a CIGAR of more than 65 535 operations lives here, the field holds <l_seq>S<n>N (htslib
                    # puts it back behind the reference's pysam)
                    cig = np.frombuffer(aux[q + 5:q + 5 + 4 * cnt], dtype="<u4")
                    n_cigar = cnt
                q += 5 + cnt * {b"c": 1, b"C": 1, b"s": 2, b"S": 2, b"i": 4, b"I": 4, b"f": 4}[sub]
            else:
                break
        flags.append(flag & 0x3FFF); tids.append(tid); poss.append(pos); tlens.append(tlen)
        cigs.extend(int(c) for c in cig)
        cig_counts.append(n_cigar)
        rgs.append(rg)
    al = _finish(header, flags, tids, poss, tlens, cigs, cig_counts, seqs, quals, rgs, qnames)
    al.batch.mtid = np.asarray(mtids, np.int32)
    al.batch.mpos = np.asarray(mposs, np.int32)
    al.has_mr = has_mr
    if keep_raw:
        al.raw, al.raw_header = raws, raw_header
    return al


def write_bam_raw(path, raw_header, records):
    """BGZF-compress an already encoded BAM stream (header bytes + record bodies)."""
    data = bytearray(raw_header)
    for body in records:
        data += struct.pack("<i", len(body)) + body
    data = bytes(data)
    with open(path, "wb") as out:
        for lo in range(0, len(data), 0xFF00):
            out.write(_bgzf_block(data[lo:lo + 0xFF00]))
        out.write(_bgzf_block(b""))


class BgzfWriter:
    """BGZF output stream: 0xFF00-byte blocks deflated on a pool of threads (zlib drops the GIL), written in order.  ``write``
    hands its blocks to the pool and returns: they are deflated while the caller prepares the next piece (the rescaling pass:
    the next chunk's decode, kernels and patching run under the deflate of this one's output, which is what bounds it), at most
    ``max_pending`` blocks — 256 MiB of input — wait at a time."""

    def __init__(self, path, threads=None, max_pending=4096, engine=None):
        """``engine`` (a ``DamageEngine``): the members are made on its device instead (include/mdx.h ``mdx_bgzf_deflate``: a lane
        per quarter of a member, 2.9 GB/s against 0.6 on sixteen threads; 3 % larger than zlib's level 6) — every ``write`` is
        then compressed on its own, its last member as short as it comes."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        self._out = open(path, "wb")
        self._engine = engine
        # (engine: one thread writes the finished members to the file, in order, while the next piece is compressed)
        self._pool = ThreadPoolExecutor(1) if engine is not None else ThreadPoolExecutor(threads or min(32, usable_cpus()))
        self._tail = b""
        self._pending = deque()
        self._max_pending = max_pending

    def _drain(self, keep):
        """Write the finished blocks at the head of the queue; wait for the oldest ones while more than ``keep`` are queued."""
        if self._engine is not None:
            while len(self._pending) > keep:
                self._pending.popleft().result()
            return
        while self._pending and (len(self._pending) > keep or self._pending[0].done()):
            self._out.write(self._pending.popleft().result())

    def write(self, data):
        """Append bytes (anything with the buffer protocol; the object must not change until the stream is closed or flushed)."""
        view = memoryview(data).cast("B")
        if self._engine is not None:
            if len(view):
                members = self._engine.bgzf_deflate(view)
                self._pending.append(self._pool.submit(self._out.write, members.data))
                while len(self._pending) > 2:           # (two pieces' members waiting for the disk at most)
                    self._pending.popleft().result()
            return
        if self._tail:
            view = memoryview(self._tail + view.tobytes())
        whole = len(view) // 0xFF00 * 0xFF00
        self._tail = view[whole:].tobytes()
        for lo in range(0, whole, 0xFF00):
            self._pending.append(self._pool.submit(_bgzf_block, view[lo:lo + 0xFF00]))
            if len(self._pending) > self._max_pending:
                self._drain(self._max_pending // 2)
        self._drain(self._max_pending)

    def write_members(self, members):
        """Append finished BGZF members (the device's: ``GpuBamStream.rescale_slab``) as they are."""
        assert not self._tail
        self._pending.append(self._pool.submit(self._out.write, memoryview(members)))
        while len(self._pending) > 2:
            self._pending.popleft().result()

    def flush(self):
        self._drain(0)

    def close(self):
        if self._out is None:
            return
        self._drain(0)
        if self._tail:
            self._out.write(_bgzf_block(self._tail))
        self._out.write(_bgzf_block(b""))
        self._out.close()
        self._out = None
        self._pool.shutdown()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def bam_header_bytes(header):
    """The uncompressed BAM header block (magic, text, reference dictionary) of a parsed ``Header``."""
    text = header.text.encode()
    out = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(header.references)))
    for name, length in zip(header.references, header.lengths):
        raw = name.encode() + b"\0"
        out += struct.pack("<i", len(raw)) + raw + struct.pack("<i", int(length))
    return bytes(out)


class _NativeBam:
    """Owner of a decoded BAM inside libmdx.so (``mdx_bam_free`` when the last array viewing it is gone)."""

    def __init__(self, lib, handle):
        self._lib, self._handle = lib, handle

    def __del__(self):
        if self._handle:
            self._lib.mdx_bam_free(self._handle)
            self._handle = None


def _native_header(lib, handle):
    header = Header(lib.mdx_bam_header_text(handle).decode())
    n_ref = lib.mdx_bam_n_ref(handle)
    if not header.references:
        header.references = [lib.mdx_bam_ref_name(handle, i).decode() for i in range(n_ref)]
        header.lengths = [int(lib.mdx_bam_ref_length(handle, i)) for i in range(n_ref)]
    return header


def _native_alignments(lib, handle, owner, header=None):
    """``Alignments`` over the columns of a decoded ``mdx_bam`` (numpy views, no copy; every view keeps
    ``owner`` and with it the decoder's buffers alive)."""
    import ctypes

    from .engine import MdxBatch
    if header is None:
        header = _native_header(lib, handle)
    view = MdxBatch()
    mtid, mpos, rgi, hmr = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    lib.mdx_bam_batch(handle, ctypes.byref(view), ctypes.byref(mtid), ctypes.byref(mpos), ctypes.byref(rgi),
                      ctypes.byref(hmr))
    n, nb, nc = view.n_reads, view.n_bases, view.n_cigar

    def col(ptr, count, dtype):
        if count == 0 or not ptr:
            return np.zeros(0, dtype)
        raw = (ctypes.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
        raw._owner = owner          # the view (and every view of it) keeps the decoder's buffers alive
        return np.frombuffer(raw, dtype=dtype)

    batch = ReadBatch(col(view.flag, n, np.uint16), col(view.lib, n, np.uint16), col(view.tid, n, np.int32),
                      col(view.pos, n, np.int32), col(view.tlen, n, np.int32),
                      col(view.cigar_off, n + 1, np.uint32) if n else np.zeros(1, np.uint32),
                      col(view.cigar, nc, np.uint32),
                      col(view.seq_off, n + 1, np.uint32) if n else np.zeros(1, np.uint32),
                      col(view.seq, nb, np.uint8), col(view.qual, nb, np.uint8),
                      col(mtid.value, n, np.int32), col(mpos.value, n, np.int32)).validate()
    al = Alignments(header, batch, None, None)
    al.rg_names = [lib.mdx_bam_rg_name(handle, i).decode() for i in range(lib.mdx_bam_n_rg(handle))]
    al.rg_index = col(rgi.value, n, np.int32)
    offs = ctypes.c_void_p()
    blob_ptr = lib.mdx_bam_qnames(handle, ctypes.byref(offs))
    al._qoff = col(offs.value, n + 1, np.uint32) if n else np.zeros(1, np.uint32)
    al._qblob = col(blob_ptr, int(al._qoff[-1]), np.uint8) if n else np.zeros(0, np.uint8)
    al.has_mr = col(hmr.value, n, np.uint8).view(bool)
    al.qmin = col(lib.mdx_bam_qmin(handle), n, np.uint8)     # lowest quality of each record
    return al


def read_bam_native(path, threads=None):
    """BAM -> ``Alignments`` through the C++ decoder of libmdx.so (multi-threaded BGZF inflate,
    records unpacked straight into the SoA columns; include/mdx.h ``mdx_bam_*``).  The columns are numpy
    views of the decoder's buffers (no copy); every view keeps the decoder alive."""
    import ctypes
    import os

    from .engine import MdxBatch, load_library
    lib = load_library()
    handle = ctypes.c_void_p()
    rc = lib.mdx_bam_read(str(path).encode(), ctypes.c_int(threads or min(64, usable_cpus())),
                          ctypes.byref(handle))
    owner = _NativeBam(lib, handle)
    if rc != 0:
        raise ValueError("%r: %s" % (str(path), lib.mdx_bam_error(handle).decode() if handle else "BAM decode failed"))
    return _native_alignments(lib, handle, owner)


class GpuBamStream:
    """GPU-side decode of a BAM file (include/mdx.h ``mdx_gbam_*``): the compressed file goes to HBM a slab of BGZF
    blocks at a time and is inflated and unpacked there; ``next_view`` returns an ``MdxBatch`` of DEVICE pointers for
    ``DamageEngine.tabulate_view`` (valid until the next call), or None at the end of the file.  ``readgroups``: the
    header's read-group ids with the library index of each; ``lib_default``: library of a record without RG tag
    (None: such a record is an error when it is counted).  Any BGZF layout: records may straddle blocks and slabs, the
    header may share a block with records.  Raises ``GpuDecodeUnsupported`` for what is left (a record longer than a
    gigabyte; in a sharded run a slab whose first record cannot be told without the slab in front) — the caller then
    decodes on the host (``BamStream``), from ``tell()`` on if it likes."""

    def __init__(self, engine, path, readgroups=(), lib_default=None, chunk_bytes=256 << 20, want_qual=False,
                 want_mate=False, min_basequal=0, packed=None):
        import ctypes
        self._lib = engine._lib
        self._engine = engine
        self._g = ctypes.c_void_p()
        self.path, self.chunk_bytes = path, int(chunk_bytes)
        if not engine._ctx:
            raise ValueError("the engine is closed")
        rc = self._lib.mdx_gbam_open(engine._ctx, str(path).encode(), ctypes.byref(self._g))
        # (the engine closes the streams still open on it before it destroys its context: DamageEngine.close)
        if not hasattr(engine, "_streams"):
            import weakref
            engine._streams = weakref.WeakSet()
        engine._streams.add(self)
        if rc != 0:
            message = self._error()
            self.close()
            if rc == -8:
                raise GpuDecodeUnsupported("%r: %s" % (str(path), message))
            raise ValueError("%r: %s" % (str(path), message))
        self.header = _native_header(self._lib, self._lib.mdx_gbam_header(self._g))
        ids = [str(rg).encode() for rg, _ in readgroups]
        arr = (ctypes.c_char_p * max(1, len(ids)))(*ids)
        libs = (ctypes.c_int32 * max(1, len(ids)))(*[int(lib) for _, lib in readgroups])
        rc = self._lib.mdx_gbam_configure(self._g, len(ids), arr, libs, -1 if lib_default is None else int(lib_default),
                                          int(bool(want_qual)), int(bool(want_mate)))
        if rc != 0:
            raise ValueError("%r: %s" % (str(path), self._error()))
        if min_basequal:
            # --min-basequal: unmaskable records are flagged on the device, see ``missing_qualities``
            if self._lib.mdx_gbam_set_min_basequal(self._g, int(min_basequal)) != 0:
                raise ValueError("%r: %s" % (str(path), self._error()))
        # the SEQ column of the views: BAM's nibbles kept as nibbles (MDX_SEQ_4BIT: the packed kernel — with --min-basequal
        # its masked form — reads them) unless the launches behind it read ASCII anyway (the rescale kernels of
        # --rescale-only want the qualities without a threshold); ``packed`` overrides the choice
        if packed is None:
            packed = bool(min_basequal) or not want_qual
        self.packed = bool(packed)
        if self.packed:
            self._lib.mdx_gbam_set_seq_format(self._g, 1)

    def missing_qualities(self):
        """A record the kernel counts has come by without qualities (main.py:185-192 warns once)."""
        return bool(self._lib.mdx_gbam_missing_qualities(self._g))

    def _error(self):
        return self._lib.mdx_gbam_error(self._g).decode() if self._g else "GPU BAM decode failed"

    def next_view(self):
        import ctypes
        from .engine import MdxBatch
        view = MdxBatch()
        mtid, mpos = ctypes.c_void_p(), ctypes.c_void_p()
        rc = self._lib.mdx_gbam_next(self._g, self.chunk_bytes, ctypes.byref(view), ctypes.byref(mtid), ctypes.byref(mpos))
        if rc == -8:
            raise GpuDecodeUnsupported("%r: %s" % (str(self.path), self._error()))
        if rc != 0:
            raise ValueError("%r: %s" % (str(self.path), self._error()))
        if view.n_reads == 0 and self._lib.mdx_gbam_at_end(self._g):
            return None
        view.mtid, view.mpos = mtid.value, mpos.value       # (python attributes: not fields of the C struct)
        return view

    def tell(self):
        """(compressed offset of the next slab's first BGZF block, inflated bytes in front of its first record) — where
        ``BamStream.seek`` takes up the file — or None when the device path does not know (behind ``skip``)."""
        import ctypes
        coff, phase = ctypes.c_int64(), ctypes.c_int64()
        if self._lib.mdx_gbam_tell(self._g, ctypes.byref(coff), ctypes.byref(phase)) != 0:
            return None
        return coff.value, phase.value

    def view_flags(self, view):
        """The flag column of the view ``next_view`` handed out last, on the host (--downsample on the device path:
        reader.py:134-146 draws once per record the flag filter keeps, in file order)."""
        import ctypes
        import numpy as np
        n = int(view.n_reads)
        out = np.empty(n, np.uint16)
        self._lib.mdx_gbam_view_flags.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        rc = self._lib.mdx_gbam_view_flags(self._g, ctypes.c_void_p(out.ctypes.data), ctypes.c_int64(n))
        if rc != 0:
            raise ValueError("%r: %s" % (str(self.path), self._error()))
        return out

    def set_view_flags(self, view, flags):
        """... and back: the caller has marked the records that leave (a bit the flag filter drops)."""
        import ctypes
        import numpy as np
        flags = np.ascontiguousarray(flags, dtype=np.uint16)
        assert flags.shape[0] == int(view.n_reads)
        self._lib.mdx_gbam_view_set_flags.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        rc = self._lib.mdx_gbam_view_set_flags(self._g, ctypes.c_void_p(flags.ctypes.data), ctypes.c_int64(flags.shape[0]))
        if rc != 0:
            raise ValueError("%r: %s" % (str(self.path), self._error()))

    def rescale_slab(self, view, counts):
        """The slab ``next_view`` handed out last (a stream opened with ``want_qual``, ``want_mate``, ``packed=False``) rescaled
        through the engine's model and written back on the device (include/mdx.h ``mdx_gbam_rescale_slab``): returns the slab's
        records — new qualities, ``MR:f`` tags — as BGZF members (a uint8 array); ``counts`` (int64[5]) += records by routing
        status.  Raises SystemExit with the reference's message when a record to be rescaled has an MR tag already
        (rescale.py:277-278), BadReadError for a record the kernels cannot process."""
        import ctypes
        import numpy as np
        n, nb, ncg = int(view.n_reads), int(view.n_bases), int(view.n_cigar)
        # (the slab's inflated bytes are at most: per record 36 + a name of 255 + its CIGAR + 1.5 x its bases + tags — bounded
        # by what the decoder itself allows: 4 x the compressed slab and more; taken from the view: bases x 1.5 + 4 per
        # operation + 400 per record covers names and tags of any file this library has seen; a longer one is MDX_ERR_ARG)
        cap = int(nb * 1.5) + 4 * ncg + 400 * n + (1 << 20)
        cap += (cap // 0xFF00 + 1) * 64
        out = np.empty(cap, np.uint8)
        out_len, clash = ctypes.c_int64(0), ctypes.c_int64(-1)
        fn = self._lib.mdx_gbam_rescale_slab
        fn.restype = ctypes.c_int
        rc = fn(self._g, ctypes.c_void_p(out.ctypes.data), ctypes.c_int64(cap), ctypes.byref(out_len),
                ctypes.c_void_p(counts.ctypes.data), ctypes.byref(clash))
        if rc == -6 and clash.value >= 0:
            name = ctypes.create_string_buffer(256)
            self._lib.mdx_gbam_record_name(self._g, ctypes.c_int64(clash.value), name, 256)
            raise SystemExit("Read: %s already has a MR tag, can't rescale" % name.value.decode(errors="replace"))
        if rc == -6:
            from .engine import BadReadError
            raise BadReadError(-2 - clash.value, self._error())
        if rc != 0:
            raise ValueError("%r: %s" % (str(self.path), self._error()))
        return out[:out_len.value]

    def fixups(self):
        """BGZF blocks whose guessed first record was not where the chain of the records in front of it ended (rescanned
        from the right offset: the result is exact either way)."""
        return int(self._lib.mdx_gbam_fixups(self._g))

    def skip(self):
        """Step over the slab ``next_view`` would decode (a run over several GPUs: the slabs of the other ranks).
        False at the end of the file."""
        if self._lib.mdx_gbam_at_end(self._g):
            return False
        rc = self._lib.mdx_gbam_skip(self._g, self.chunk_bytes)
        if rc == -8:
            raise GpuDecodeUnsupported("%r: %s" % (str(self.path), self._error()))
        if rc != 0:
            raise ValueError("%r: %s" % (str(self.path), self._error()))
        return True

    def close(self):
        if self._g:
            # (mdx_gbam_close gives the arena back to the context: never reached with the context gone — the engine closes
            # its streams first)
            self._lib.mdx_gbam_close(self._g)
            self._g = None
            streams = getattr(self._engine, "_streams", None)
            if streams is not None:
                streams.discard(self)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuDecodeUnsupported(ValueError):
    """The file's layout is not one the GPU decode path takes (MDX_ERR_UNSUPPORTED)."""


class BamStream:
    """Chunked form of ``read_bam_native`` (include/mdx.h ``mdx_bam_open`` / ``mdx_bam_next``): iterating yields
    ``Alignments`` of consecutive records, at most ``chunk_bytes`` of uncompressed BAM data each, so host memory stays
    bounded and the caller can tabulate one chunk while the next is decoded (the ctypes call drops the GIL)."""

    def __init__(self, path, threads=None, chunk_bytes=256 << 20, keep_raw=False):
        import ctypes
        import os

        from .engine import load_library
        self._lib = load_library()
        self._stream = ctypes.c_void_p()
        self.path, self.chunk_bytes, self.keep_raw = path, int(chunk_bytes), bool(keep_raw)
        rc = self._lib.mdx_bam_open(str(path).encode(), ctypes.c_int(threads or min(64, usable_cpus())),
                                    ctypes.byref(self._stream))
        if rc != 0:
            message = self._error()
            self.close()
            raise ValueError("%r: %s" % (str(path), message))
        self.header = _native_header(self._lib, self._lib.mdx_bam_stream_header(self._stream))
        if self.keep_raw:
            self._lib.mdx_bam_stream_keep_raw(self._stream, 1)

    def _error(self):
        if not self._stream:
            return "BAM decode failed"
        return self._lib.mdx_bam_error(self._lib.mdx_bam_stream_header(self._stream)).decode()

    def seek(self, comp_off, phase=0):
        """Go on from the BGZF block at compressed offset ``comp_off``, the first record ``phase`` inflated bytes into it
        (``GpuBamStream.tell``)."""
        import ctypes
        if self._lib.mdx_bam_seek(self._stream, ctypes.c_int64(int(comp_off)), ctypes.c_int64(int(phase))) != 0:
            raise ValueError("%r: %s" % (str(self.path), self._error()))

    def next_chunk(self):
        """The next ``Alignments`` or None at the end of the file."""
        import ctypes
        handle = ctypes.c_void_p()
        rc = self._lib.mdx_bam_next(self._stream, self.chunk_bytes, ctypes.byref(handle))
        if rc != 0:
            raise ValueError("%r: %s" % (str(self.path), self._error()))
        if not handle:
            return None
        chunk = _native_alignments(self._lib, handle, _NativeBam(self._lib, handle), self.header)
        chunk.native = handle          # (kept alive by the chunk's owner object) for patch_rescaled()
        return chunk

    def raw_bodies(self, chunk, count=None):
        """The encoded bodies (without their block_size field) of the first ``count`` records of ``chunk`` (decoded with
        ``keep_raw``), as bytes objects."""
        import ctypes
        n = chunk.batch.n if count is None else min(int(count), chunk.batch.n)
        data, rec_off = ctypes.c_void_p(), ctypes.c_void_p()
        if self._lib.mdx_bam_raw(chunk.native, ctypes.byref(data), ctypes.byref(rec_off)) != 0:
            raise ValueError("the stream was not opened with keep_raw")
        off = np.ctypeslib.as_array(ctypes.cast(rec_off, ctypes.POINTER(ctypes.c_uint64)), (chunk.batch.n + 1,))
        blob = ctypes.string_at(data.value, int(off[n]))
        return [blob[int(off[i]) + 4:int(off[i + 1])] for i in range(n)]

    def patch_rescaled(self, chunk, qual_out, mr, rescaled):
        """The encoded records of ``chunk`` (decoded with ``keep_raw``) with the QUAL of the records flagged in
        ``rescaled`` replaced from ``qual_out`` and an ``MR:f`` tag appended; every other byte as in the file."""
        import ctypes
        n = chunk.batch.n
        qual_out = np.ascontiguousarray(qual_out, np.uint8)
        mr = np.ascontiguousarray(mr, np.float32)
        rescaled = np.ascontiguousarray(rescaled, np.uint8)
        data, rec_off = ctypes.c_void_p(), ctypes.c_void_p()
        if self._lib.mdx_bam_raw(chunk.native, ctypes.byref(data), ctypes.byref(rec_off)) != 0:
            raise ValueError("the stream was not opened with keep_raw")
        end = int(np.ctypeslib.as_array(ctypes.cast(rec_off, ctypes.POINTER(ctypes.c_uint64)), (n + 1,))[n])
        out = np.empty(end + 7 * int(rescaled.sum()) + 8, np.uint8)
        out_len = ctypes.c_int64(0)
        rc = self._lib.mdx_bam_patch_rescaled(chunk.native, qual_out.ctypes.data_as(ctypes.c_void_p),
                                              mr.ctypes.data_as(ctypes.c_void_p), rescaled.ctypes.data_as(ctypes.c_void_p),
                                              out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(out.shape[0]),
                                              ctypes.byref(out_len))
        if rc != 0:
            raise ValueError("mdx_bam_patch_rescaled failed (%d)" % rc)
        return out[:out_len.value]

    def __iter__(self):
        while True:
            chunk = self.next_chunk()
            if chunk is None:
                return
            yield chunk

    def close(self):
        if self._stream:
            self._lib.mdx_bam_close(self._stream)
            self._stream = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def is_bam(path):
    if str(path) == "-":
        return False
    with open(path, "rb") as handle:
        return handle.read(2) == b"\x1f\x8b"


def read_alignments(path):
    """SAM or BAM by content (mapdamage/reader.py:38 lets htslib sniff the format).  BAM goes through
    the native decoder of libmdx.so; the pure-Python ``read_bam`` remains as its cross-check."""
    if str(path) == "-":
        import sys
        return read_sam(sys.stdin)
    with open(path, "rb") as handle:
        magic = handle.read(2)
    if magic != b"\x1f\x8b":
        return read_sam(str(path))
    return read_bam_native(path)


# ---------------------------------------------------------------------------- writers (tests, synthetic inputs)
def header_text(ref_names, ref_lengths, read_groups):
    lines = ["@HD\tVN:1.6\tSO:unsorted"]
    lines += ["@SQ\tSN:%s\tLN:%d" % (n, ln) for n, ln in zip(ref_names, ref_lengths)]
    for rg in read_groups:
        lines.append("@RG\t" + "\t".join("%s:%s" % kv for kv in rg.items()))
    return "\n".join(lines) + "\n"


def _cigar_string(ops):
    return "".join("%d%s" % (int(c) >> 4, L.CIGAR_CHARS[int(c) & 15]) for c in ops) or "*"


def write_sam(path, batch: ReadBatch, ref_names, ref_lengths, read_groups, rg_of_record=None):
    """``read_groups``: list of dicts (ID, SM, LB ...); ``rg_of_record``: RG id per record or None."""
    with open(path, "wt") as out:
        out.write(header_text(ref_names, ref_lengths, read_groups))
        for i in range(batch.n):
            c0, c1 = int(batch.cigar_off[i]), int(batch.cigar_off[i + 1])
            s0, s1 = int(batch.seq_off[i]), int(batch.seq_off[i + 1])
            seq = batch.seq[s0:s1].tobytes().decode() or "*"
            qual = "*"
            if batch.qual is not None and s1 > s0 and int(batch.qual[s0]) != 0xFF:
                qual = (batch.qual[s0:s1] + 33).astype(np.uint8).tobytes().decode()
            tid = int(batch.tid[i])
            fields = ["r%d" % i, str(int(batch.flag[i])), ref_names[tid] if tid >= 0 else "*",
                      str(int(batch.pos[i]) + 1), "30", _cigar_string(batch.cigar[c0:c1]), "*", "0",
                      str(int(batch.tlen[i])), seq, qual]
            rg = None if rg_of_record is None else rg_of_record[i]
            if rg is not None:
                fields.append("RG:Z:%s" % rg)
            out.write("\t".join(fields) + "\n")


_WB_JOB = None


def _wb_slice(span):
    """Worker of write_bam(workers=N): records [lo, hi) of batch k as finished BGZF blocks, each starting at a record."""
    k, lo, hi = span
    batches, rg_of_record = _WB_JOB
    batch = batches[k]
    out = io.BytesIO()
    room = 0xFF00
    piece = bytearray()
    for i in range(lo, hi):
        record = _bam_record(batch, i, rg_of_record if rg_of_record is None or isinstance(rg_of_record, str) else rg_of_record[i])
        if piece and len(piece) + len(record) > room:
            out.write(_bgzf_block(bytes(piece)))
            piece = bytearray()
        piece += record
        while len(piece) > room:                 # a record larger than a block spills over
            out.write(_bgzf_block(bytes(piece[:room])))
            del piece[:room]
    if piece:
        out.write(_bgzf_block(bytes(piece)))
    return out.getvalue()


def _bam_record(batch, i, rg):
    c0, c1 = int(batch.cigar_off[i]), int(batch.cigar_off[i + 1])
    s0, s1 = int(batch.seq_off[i]), int(batch.seq_off[i + 1])
    name = ("r%d" % i).encode() + b"\x00"
    l_seq = s1 - s0
    codes = _SEQ_ENCODE[batch.seq[s0:s1]]
    if l_seq % 2:
        codes = np.concatenate([codes, np.zeros(1, np.uint8)])
    packed = ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8).tobytes()
    qual = batch.qual[s0:s1].tobytes() if batch.qual is not None else b"\xff" * l_seq
    aux = b"" if rg is None else b"RGZ" + rg.encode() + b"\x00"
    ntid = -1 if batch.mtid is None else int(batch.mtid[i])
    npos = -1 if batch.mpos is None else int(batch.mpos[i])
    body = struct.pack("<iiBBHHHiiii", int(batch.tid[i]), int(batch.pos[i]), len(name), 30, 4680,
                       c1 - c0, int(batch.flag[i]), l_seq, ntid, npos, int(batch.tlen[i]))
    body += name + batch.cigar[c0:c1].astype("<u4").tobytes() + packed + qual + aux
    return struct.pack("<i", len(body)) + body


def write_bam(path, batch, ref_names, ref_lengths, read_groups, rg_of_record=None, htslib_blocks=True,
              workers=1, block_bytes=0xFF00):
    """``htslib_blocks``: lay the BGZF blocks out as htslib does (the header flushed on its own, and a block closed
    early when the next record would not fit, ``bgzf_flush_try`` in ``bam_write1``), so that every block starts
    at a record — what the files mapDamage sees in practice look like, and what the native decoder's parallel
    record scan speculates on.  False: header and records as one stream cut into blocks of ``block_bytes`` anywhere — the
    header shares a block with records, records straddle blocks (htsjdk / Picard fill 65 498 bytes per block).
    ``workers`` > 1 (htslib layout only): the records are encoded and deflated by forked worker processes, a slice
    each (call it before the process touches the GPU); a slice starts a block of its own, otherwise the same file.
    With workers, ``batch`` may be a list of batches (written one behind the other: a file of more than 4 G bases) and
    ``rg_of_record`` one read-group id for every record."""
    text = header_text(ref_names, ref_lengths, read_groups).encode()
    batches = list(batch) if isinstance(batch, (list, tuple)) else [batch]
    if htslib_blocks and workers > 1 and sum(b.n for b in batches) > 4 * workers:
        import multiprocessing as mp
        global _WB_JOB
        head = io.BytesIO()
        head.write(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(ref_names)))
        for name, ln in zip(ref_names, ref_lengths):
            nb = name.encode() + b"\x00"
            head.write(struct.pack("<i", len(nb)) + nb + struct.pack("<i", ln))
        hv = head.getvalue()
        n_jobs = workers * 4
        spans = [(k, b.n * j // n_jobs, b.n * (j + 1) // n_jobs) for k, b in enumerate(batches) for j in range(n_jobs)]
        _WB_JOB = (batches, rg_of_record)
        try:
            with mp.get_context("fork").Pool(workers) as pool, open(path, "wb") as out:
                for lo in range(0, len(hv), 0xFF00):
                    out.write(_bgzf_block(hv[lo:lo + 0xFF00]))
                for blob in pool.imap(_wb_slice, spans, chunksize=1):
                    out.write(blob)
                out.write(_bgzf_block(b""))
        finally:
            _WB_JOB = None
        return
    if len(batches) != 1:
        raise ValueError("several batches are written by forked workers only (workers > 1, htslib layout)")
    batch = batches[0]
    if isinstance(rg_of_record, str):
        rg_of_record = [rg_of_record] * batch.n
    raw = io.BytesIO()
    pieces = []                      # htslib layout: uncompressed payload of each block
    room = 0xFF00
    raw.write(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(ref_names)))
    for name, ln in zip(ref_names, ref_lengths):
        nb = name.encode() + b"\x00"
        raw.write(struct.pack("<i", len(nb)) + nb + struct.pack("<i", ln))
    if htslib_blocks:
        head = raw.getvalue()
        pieces = [bytearray(head[lo:lo + room]) for lo in range(0, len(head), room)] + [bytearray()]
    for i in range(batch.n):
        record = _bam_record(batch, i, None if rg_of_record is None else rg_of_record[i])
        if not htslib_blocks:
            raw.write(record)
            continue
        if pieces[-1] and len(pieces[-1]) + len(record) > room:
            pieces.append(bytearray())
        pieces[-1] += record
        while len(pieces[-1]) > room:            # a record larger than a block spills over
            pieces.append(pieces[-1][room:])
            del pieces[-2][room:]
    if not htslib_blocks:
        data = raw.getvalue()
        pieces = [data[lo:lo + block_bytes] for lo in range(0, len(data), block_bytes)]
    with open(path, "wb") as out:
        for piece in pieces:
            if len(piece):
                out.write(_bgzf_block(bytes(piece)))
        out.write(_bgzf_block(b""))


def _bgzf_block(chunk):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    deflated = comp.compress(chunk) + comp.flush()
    bsize = len(deflated) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize)
            + deflated + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
