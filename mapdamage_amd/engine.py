"""ctypes binding of libmdx.so (include/mdx.h) — the product path.

There is no CPU fallback here: if the HIP library is missing or no GPU is visible the
constructor raises.  ``DamageEngine`` replaces the three accumulator objects the reference
creates in mapdamage/main.py:147-155 and the loop body main.py:165-217 that feeds them.
"""

import ctypes
import os
import pathlib

import numpy as np

from . import layout as L
from .batch import ReadBatch, Reference
from .tables import TableSet

_HERE = pathlib.Path(__file__).resolve().parent
_LIBPATH = _HERE / "libmdx.so"
_lib = None

EXPORTS = (
    "mdx_abi_version", "mdx_strerror", "mdx_create", "mdx_destroy", "mdx_last_error",
    "mdx_set_stream", "mdx_set_reference", "mdx_batch_upload", "mdx_batch_free",
    "mdx_tabulate_host", "mdx_tabulate_device", "mdx_sync", "mdx_table_words",
    "mdx_finish_device", "mdx_finish", "mdx_reset", "mdx_timing_enable", "mdx_timing_read",
    "mdx_table_mode", "mdx_genome_composition", "mdx_rescale_set_model", "mdx_rescale_host",
    "mdx_rescale_summary_words", "mdx_rescale_summary",
    "mdx_comm_unique_id", "mdx_comm_init", "mdx_comm_adopt", "mdx_comm_size", "mdx_finish_allreduce",
    "mdx_rescale_device", "mdx_tabulate_rescale_device", "mdx_rescale_timing_read", "mdx_fused_launches",
    "mdx_bam_read", "mdx_bam_free", "mdx_bam_error", "mdx_bam_header_text", "mdx_bam_n_ref", "mdx_bam_ref_name",
    "mdx_bam_ref_length", "mdx_bam_batch", "mdx_bam_n_rg", "mdx_bam_rg_name", "mdx_bam_qnames",
    "mdx_bam_open", "mdx_bam_stream_header", "mdx_bam_next", "mdx_bam_close",
    "mdx_bam_stream_keep_raw", "mdx_bam_raw", "mdx_bam_patch_rescaled", "mdx_bam_qmin",
    "mdx_ctx_stream", "mdx_gbam_open", "mdx_gbam_header", "mdx_gbam_error", "mdx_gbam_configure", "mdx_gbam_next",
    "mdx_gbam_at_end", "mdx_gbam_close", "mdx_gbam_set_min_basequal", "mdx_gbam_missing_qualities",
    "mdx_gbam_inflate_blocks", "mdx_set_record_base", "mdx_pack_seq", "mdx_gbam_set_seq_format", "mdx_packed_launches",
    "mdx_gbam_skip", "mdx_comm_count", "mdx_gbam_tell", "mdx_gbam_fixups", "mdx_bam_seek", "mdx_libsorts",
    "mdx_gbam_view_flags", "mdx_gbam_view_set_flags",
    "mdx_rescale_patches_device", "mdx_tabulate_rescale_patches_device", "mdx_rescale_expand_device", "mdx_mr_round", "mdx_batch_fold", "mdx_bgzf_deflate",
    "mdx_gbam_rescale_slab", "mdx_gbam_write_rescaled", "mdx_gbam_record_name",
    "mdx_fasta_index", "mdx_set_reference_fasta", "mdx_reference_fetch", "mdx_host_threads", "mdx_host_pool_threads", "mdx_warm",
)

SEQ_ASCII, SEQ_4BIT, SEQ_4BITQ = 0, 1, 2      # include/mdx.h MDX_SEQ_*


class MdxConfig(ctypes.Structure):
    _fields_ = [("length", ctypes.c_int32), ("around", ctypes.c_int32),
                ("minqual", ctypes.c_int32), ("nlib", ctypes.c_int32),
                ("lgd_max", ctypes.c_int32), ("device", ctypes.c_int32),
                ("lgd_over_cap", ctypes.c_int64)]


class MdxBatch(ctypes.Structure):
    _fields_ = [("n_reads", ctypes.c_int64), ("n_cigar", ctypes.c_int64),
                ("n_bases", ctypes.c_int64),
                ("flag", ctypes.c_void_p), ("lib", ctypes.c_void_p), ("tid", ctypes.c_void_p),
                ("pos", ctypes.c_void_p), ("tlen", ctypes.c_void_p),
                ("cigar_off", ctypes.c_void_p), ("cigar", ctypes.c_void_p),
                ("seq_off", ctypes.c_void_p), ("seq", ctypes.c_void_p),
                ("qual", ctypes.c_void_p),
                ("seq_format", ctypes.c_int32), ("reserved", ctypes.c_int32), ("lowq", ctypes.c_void_p),
                ("libsort", ctypes.c_void_p)]


class MdxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libmdx error %d: %s" % (code, message))
        self.code = code


class BadReadError(ValueError):
    """A record the reference itself cannot process, e.g. an alignment running past the end
    of its contig, where pysam's ``FastaFile.fetch`` raises ``ValueError`` (align.py:33)."""

    def __init__(self, read_index, message):
        super().__init__("invalid coordinates or record at batch index %d: %s"
                         % (read_index, message))
        self.read_index = read_index


def load_library(path=None):
    """Load libmdx.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # (MDX_LIBPATH: another build of the library — the A/B variants of tools/mkvariant.sh)
    p = pathlib.Path(path) if path else pathlib.Path(os.environ.get("MDX_LIBPATH") or _LIBPATH)
    if not p.exists():
        raise RuntimeError("%s is missing: build it with `python -m mapdamage_amd.build` "
                           "(hipcc, gfx950); there is no CPU fallback" % p)
    lib = ctypes.CDLL(str(p))
    lib.mdx_strerror.restype = ctypes.c_char_p
    lib.mdx_last_error.restype = ctypes.c_char_p
    lib.mdx_last_error.argtypes = [ctypes.c_void_p]
    lib.mdx_table_words.restype = ctypes.c_int64
    lib.mdx_table_words.argtypes = [ctypes.c_void_p]
    lib.mdx_destroy.restype = None
    lib.mdx_destroy.argtypes = [ctypes.c_void_p]
    for name in ("mdx_set_stream", "mdx_set_reference", "mdx_batch_upload", "mdx_batch_free",
                 "mdx_tabulate_host", "mdx_tabulate_device", "mdx_sync", "mdx_finish_device",
                 "mdx_finish", "mdx_reset", "mdx_timing_enable", "mdx_timing_read",
                 "mdx_table_mode", "mdx_genome_composition", "mdx_rescale_set_model", "mdx_rescale_host",
                 "mdx_rescale_summary", "mdx_comm_unique_id", "mdx_comm_init", "mdx_comm_adopt", "mdx_comm_size",
                 "mdx_finish_allreduce", "mdx_rescale_device", "mdx_tabulate_rescale_device", "mdx_rescale_timing_read",
                 "mdx_set_record_base"):
        getattr(lib, name).restype = ctypes.c_int
    lib.mdx_comm_size.argtypes = [ctypes.c_void_p]
    lib.mdx_comm_count.argtypes = [ctypes.c_void_p]
    lib.mdx_comm_adopt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
    lib.mdx_rescale_summary_words.restype = ctypes.c_int64
    lib.mdx_rescale_summary_words.argtypes = [ctypes.c_void_p]
    lib.mdx_fused_launches.restype = ctypes.c_int64
    lib.mdx_fused_launches.argtypes = [ctypes.c_void_p]
    lib.mdx_packed_launches.restype = ctypes.c_int64
    lib.mdx_packed_launches.argtypes = [ctypes.c_void_p]
    lib.mdx_libsorts.restype = ctypes.c_int64
    lib.mdx_libsorts.argtypes = [ctypes.c_void_p]
    for name in ("mdx_bam_error", "mdx_bam_header_text", "mdx_bam_ref_name", "mdx_bam_rg_name"):
        getattr(lib, name).restype = ctypes.c_char_p
    lib.mdx_bam_qnames.restype = ctypes.c_void_p
    lib.mdx_bam_qmin.restype = ctypes.c_void_p
    lib.mdx_bam_qmin.argtypes = [ctypes.c_void_p]
    lib.mdx_bam_ref_length.restype = ctypes.c_int64
    lib.mdx_bam_free.restype = None
    for name in ("mdx_bam_error", "mdx_bam_header_text", "mdx_bam_n_ref", "mdx_bam_n_rg", "mdx_bam_free"):
        getattr(lib, name).argtypes = [ctypes.c_void_p]
    for name in ("mdx_bam_ref_name", "mdx_bam_ref_length", "mdx_bam_rg_name"):
        getattr(lib, name).argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.mdx_bam_stream_header.restype = ctypes.c_void_p
    lib.mdx_bam_stream_header.argtypes = [ctypes.c_void_p]
    lib.mdx_bam_next.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    lib.mdx_bam_close.restype = None
    lib.mdx_bam_close.argtypes = [ctypes.c_void_p]
    lib.mdx_gbam_open.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p]
    lib.mdx_gbam_header.restype = ctypes.c_void_p
    lib.mdx_gbam_header.argtypes = [ctypes.c_void_p]
    lib.mdx_gbam_error.restype = ctypes.c_char_p
    lib.mdx_gbam_error.argtypes = [ctypes.c_void_p]
    lib.mdx_gbam_configure.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                       ctypes.c_int, ctypes.c_int]
    lib.mdx_gbam_next.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.mdx_gbam_at_end.argtypes = [ctypes.c_void_p]
    lib.mdx_gbam_skip.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    lib.mdx_gbam_tell.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.mdx_gbam_fixups.argtypes = [ctypes.c_void_p]
    lib.mdx_bam_seek.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
    lib.mdx_gbam_set_min_basequal.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.mdx_gbam_set_seq_format.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.mdx_pack_seq.restype = ctypes.c_int
    lib.mdx_pack_seq.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32]
    lib.mdx_gbam_missing_qualities.argtypes = [ctypes.c_void_p]
    lib.mdx_gbam_close.restype = None
    lib.mdx_gbam_close.argtypes = [ctypes.c_void_p]
    lib.mdx_fasta_index.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int32]
    lib.mdx_set_reference_fasta.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                            ctypes.c_void_p]
    lib.mdx_reference_fetch.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    lib.mdx_warm.argtypes = [ctypes.c_int32, ctypes.c_int64]
    if path is None:
        _lib = lib
    return lib


def _ptr(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def pack_seq(seq, threads=None):
    """ASCII SEQ bytes -> the 4-bit column of ``MDX_SEQ_4BIT`` (include/mdx.h): two bases per byte, low nibble first,
    1 = A, 2 = C, 4 = T, 8 = G, 0 = anything else — all the reference's loop distinguishes (statistics.py:27, 101)."""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    out = np.empty((seq.shape[0] + 1) // 2, np.uint8)
    if threads is None:
        # (the control group's quota, not the hardware's thread count: mdx_pack_seq with 0 would start one thread per hardware thread)
        from .sam import usable_cpus
        threads = min(64, usable_cpus())
    rc = load_library().mdx_pack_seq(_ptr(seq), ctypes.c_int64(seq.shape[0]), _ptr(out), ctypes.c_int32(threads))
    if rc != 0:
        raise MdxError(rc, "mdx_pack_seq")
    return out


def _host_batch(batch: ReadBatch, packed=False):
    b = MdxBatch()
    b.n_reads = batch.n
    b.n_cigar = int(batch.cigar.shape[0])
    b.n_bases = int(batch.seq.shape[0])
    keep = []
    for name in ("flag", "lib", "tid", "pos", "tlen", "cigar_off", "cigar", "seq_off", "seq"):
        arr = np.ascontiguousarray(getattr(batch, name))
        if name == "seq" and packed:
            arr = pack_seq(arr)
            b.seq_format = SEQ_4BIT
        setattr(b, name, arr.ctypes.data)
        keep.append(arr)
    if batch.qual is not None:
        q = np.ascontiguousarray(batch.qual)
        b.qual = q.ctypes.data
        keep.append(q)
    b._keep = keep  # the columns must outlive the call
    return b


class DeviceBatch:
    """A batch resident in HBM (allocated by the library)."""

    def __init__(self, engine, dev, n_reads, n_bases, n_cigar):
        self._engine = engine
        self.dev = dev
        self.n = n_reads
        self.n_bases = n_bases
        self.n_cigar = n_cigar

    def free(self):
        if self.dev is not None and self._engine._ctx:
            self._engine._lib.mdx_batch_free(self._engine._ctx, ctypes.byref(self.dev))
        self.dev = None


class DamageEngine:
    """Device-side counterpart of MisincorporationRates + DNAComposition + FragmentLengths.

    ``libraries``: list of (sample, library) tuples indexed by the ``lib`` column (unique
    libraries in header order, or ``[("*", "*")]`` for --merge-libraries)."""

    # the form in which ``upload`` / ``tabulate`` hand a host batch's SEQ column over when the caller does not say
    # (False: ASCII as it stands, True: packed to 4 bits first); the tests run every parity case through both
    # (MDX_SEQ_4BIT=1 in the environment: packed, for the profiling scripts under tools/)
    default_packed = os.environ.get("MDX_SEQ_4BIT", "") == "1"

    def __init__(self, libraries, length=70, around=10, minqual=0, lgd_max=65536, device=0,
                 lgd_over_cap=1 << 20):
        self._lib = load_library()
        self.libraries = [tuple(x) for x in libraries]
        self.length, self.around, self.minqual, self.lgd_max = length, around, minqual, lgd_max
        self.lgd_over_cap = lgd_over_cap
        cfg = MdxConfig(length, around, minqual, len(self.libraries), lgd_max, device, lgd_over_cap)
        ctx = ctypes.c_void_p()
        rc = self._lib.mdx_create(ctypes.byref(cfg), ctypes.byref(ctx))
        self._ctx = ctx if ctx.value else None
        if rc != 0:
            msg = self._lib.mdx_strerror(rc).decode()
            if self._ctx:
                msg += ": " + self._lib.mdx_last_error(self._ctx).decode()
                self._lib.mdx_destroy(self._ctx)
                self._ctx = None
            raise MdxError(rc, msg + " (a HIP device is required; there is no CPU fallback)")

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != 0:
            raise MdxError(rc, self._lib.mdx_last_error(self._ctx).decode()
                           or self._lib.mdx_strerror(rc).decode())

    def close(self):
        if self._ctx:
            # (a device decode stream hands its arena back to the context when it closes — mdx_gbam_close — so the streams
            # opened on this engine are closed before the context goes)
            for stream in list(getattr(self, "_streams", ())):
                stream.close()
            self._lib.mdx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def table_mode(self):
        return "lds" if self._lib.mdx_table_mode(self._ctx) == 0 else "global"

    def set_stream(self, hip_stream):
        self._check(self._lib.mdx_set_stream(self._ctx, ctypes.c_void_p(hip_stream or 0)))

    # ------------------------------------------------------------------ inputs
    def set_reference(self, ref):
        """``ref``: a ``Reference`` (contigs in host memory) or a ``fasta.FastaOnDisk`` — the FASTA file itself, which the
        library sends to HBM as it lies on disk and strips of its line ends there (``mdx_set_reference_fasta``)."""
        path = getattr(ref, "path", None)
        if path is not None:
            names = [n.encode() for n in ref.names]
            arr = (ctypes.c_char_p * max(1, len(names)))(*names)
            lengths = np.zeros(max(1, len(names)), np.int64)
            self._check(self._lib.mdx_set_reference_fasta(self._ctx, str(path).encode(), ctypes.c_int32(len(names)), arr,
                                                          ctypes.c_int32(1 if ref.missing_ok else 0), _ptr(lengths)))
            ref.lengths = [int(x) for x in lengths[:len(names)]]
            return
        bases, offs = ref.concat()
        self._check(self._lib.mdx_set_reference(self._ctx, _ptr(bases), _ptr(offs),
                                                ctypes.c_int32(len(ref.names))))

    def reference_fetch(self, tid, start, end):
        """``ref.fetch(chrom, start, end).upper()`` (main.py:180) as the kernels see it: the four bases, '-', and 'N' for
        every other symbol (tests)."""
        out = np.zeros(max(0, end - start), np.uint8)
        self._check(self._lib.mdx_reference_fetch(self._ctx, ctypes.c_int32(tid), ctypes.c_int64(start), ctypes.c_int64(end),
                                                  _ptr(out)))
        return out.tobytes()

    def upload(self, batch: ReadBatch, packed=None) -> DeviceBatch:
        """Resident copy of a host batch; ``packed``: with the SEQ column in its 4-bit form (``pack_seq``), which the
        tabulation (plain, with ``--min-basequal`` — the mask is folded into the column here, ``MDX_SEQ_4BITQ`` —
        or with the rescaling fused in) then reads through the packed kernels."""
        hb = _host_batch(batch, self.default_packed if packed is None else packed)
        dev = MdxBatch()
        self._check(self._lib.mdx_batch_upload(self._ctx, ctypes.byref(hb), ctypes.byref(dev)))
        return DeviceBatch(self, dev, batch.n, int(batch.seq.shape[0]), int(batch.cigar.shape[0]))

    def bgzf_deflate(self, data):
        """``data`` (bytes-like, host) as BGZF members of 0xFF00 input bytes each, deflated on the device (include/mdx.h
        ``mdx_bgzf_deflate``) -> uint8 array; no end-of-file marker."""
        view = np.frombuffer(memoryview(data).cast("B"), np.uint8)
        n = int(view.shape[0])
        out = np.empty(n + (n // 0xFF00 + 1) * 64, np.uint8)      # (a member of stored pieces: 26 + 5 per piece more than its bytes)
        out_len = ctypes.c_int64(0)
        fn = self._lib.mdx_bgzf_deflate
        fn.restype = ctypes.c_int
        self._check(fn(self._ctx, ctypes.c_void_p(view.ctypes.data), ctypes.c_int64(n), ctypes.c_void_p(out.ctypes.data),
                       ctypes.c_int64(out.shape[0]), ctypes.byref(out_len)))
        return out[:out_len.value]

    def fold(self, dbatch):
        """--min-basequal folded into a device batch's own 4-bit SEQ column, once and in place (include/mdx.h
        ``mdx_batch_fold``): a ``DeviceBatch`` or an ``MdxBatch`` of device pointers; it is ``MDX_SEQ_4BITQ`` afterwards."""
        dev = dbatch.dev if isinstance(dbatch, DeviceBatch) else dbatch
        self._lib.mdx_batch_fold.restype = ctypes.c_int
        self._check(self._lib.mdx_batch_fold(self._ctx, ctypes.byref(dev)))

    def _set_record_base(self, base):
        # (context state: every entry point that launches sets it, so that a base given to one call does not shift the
        # record numbers of the next)
        if base != getattr(self, "_record_base", 0):
            self._check(self._lib.mdx_set_record_base(self._ctx, ctypes.c_int64(base)))
            self._record_base = base

    def tabulate(self, batch, sync=True, record_base=None, packed=None):
        """Accumulate one batch (host ``ReadBatch`` or resident ``DeviceBatch``).  A host batch is staged through the
        library's pinned buffers: when the call returns its columns may be released.  ``sync=False`` leaves copies and
        kernel in flight (the next host batch is copied meanwhile); a record the reference cannot process then
        surfaces at the next ``sync()`` / ``finish()`` with index ``record_base`` + its index within its batch.
        ``packed`` (host batches): hand the SEQ column over in its 4-bit form (half the bytes across PCIe)."""
        self._set_record_base(0 if record_base is None else int(record_base))
        if isinstance(batch, DeviceBatch):
            self._check(self._lib.mdx_tabulate_device(self._ctx, ctypes.byref(batch.dev)))
        else:
            hb = _host_batch(batch, self.default_packed if packed is None else packed)
            self._check(self._lib.mdx_tabulate_host(self._ctx, ctypes.byref(hb)))
            if sync:
                # a record the reference cannot process surfaces here, with its index within this batch
                self.sync()

    def tabulate_view(self, view, record_base=None):
        """An ``MdxBatch`` of device pointers (``sam.GpuBamStream.next_view``): enqueued, errors at ``sync``."""
        self._set_record_base(0 if record_base is None else int(record_base))
        self._check(self._lib.mdx_tabulate_device(self._ctx, ctypes.byref(view)))

    def tabulate_pointers(self, n_reads, n_cigar, n_bases, **ptrs):
        """Device pointers owned by the caller (e.g. torch tensors): zero-copy entry."""
        b = MdxBatch()
        b.n_reads, b.n_cigar, b.n_bases = n_reads, n_cigar, n_bases
        for k, v in ptrs.items():
            setattr(b, k, v)
        self._set_record_base(0)
        self._check(self._lib.mdx_tabulate_device(self._ctx, ctypes.byref(b)))

    def sync(self):
        bad = ctypes.c_int64(-1)
        rc = self._lib.mdx_sync(self._ctx, ctypes.byref(bad))
        if rc == L.MDX_ERR_BAD_READ:
            raise BadReadError(bad.value, self._lib.mdx_last_error(self._ctx).decode())
        self._check(rc)

    # ------------------------------------------------------------------ outputs
    def table_words(self):
        return int(self._lib.mdx_table_words(self._ctx))

    def finish_device(self, device_ptr):
        """Write the packed canonical tables to a device buffer (for RCCL all-reduce)."""
        self._check(self._lib.mdx_finish_device(self._ctx, ctypes.c_void_p(device_ptr)))

    # ------------------------------------------------------------------ cross-device reduction (RCCL in the ABI)
    @staticmethod
    def comm_unique_id() -> bytes:
        """Rendezvous id for ``comm_init`` (ncclGetUniqueId): rank 0 creates it, every rank receives a copy."""
        buf = (ctypes.c_uint8 * 128)()
        rc = load_library().mdx_comm_unique_id(buf)
        if rc != 0:
            raise MdxError(rc, "mdx_comm_unique_id: librccl.so.1 could not be loaded")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, nranks: int, rank: int):
        """Join the communicator (collective).  Afterwards ``finish()`` returns the totals over all ranks."""
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._lib.mdx_comm_init(self._ctx, buf, ctypes.c_int32(nranks), ctypes.c_int32(rank)))

    def comm_adopt(self, rccl_comm: int, nranks: int, rank: int):
        """Use a communicator the caller owns (an ``ncclComm_t`` as an integer, e.g. torch's
        ``ProcessGroupNCCL._comm_ptr()``); it is not destroyed with the engine."""
        self._check(self._lib.mdx_comm_adopt(self._ctx, ctypes.c_void_p(rccl_comm), ctypes.c_int32(nranks), ctypes.c_int32(rank)))

    @property
    def comm_size(self):
        return int(self._lib.mdx_comm_size(self._ctx))

    def comm_count(self):
        """The ranks RCCL counts in the attached communicator (ncclCommCount); 0 without one."""
        n = int(self._lib.mdx_comm_count(self._ctx))
        if n < 0:
            self._check(n)
        return n

    def finish_allreduce(self, device_ptr):
        """finish_device + in-place ncclAllReduce(uint64, sum) on the context's stream (collective)."""
        self._check(self._lib.mdx_finish_allreduce(self._ctx, ctypes.c_void_p(device_ptr)))

    def unpack_tables(self, words: np.ndarray, lgd_over=None) -> TableSet:
        """Split a packed table block (host copy of ``finish_device`` output)."""
        from .tables import unpack_words
        return unpack_words(words, self.libraries, self.length, self.around, self.lgd_max, lgd_over)

    def finish(self) -> TableSet:
        """Synchronise and fetch the canonical tables (main.py:229-231 reads them next).  With a communicator
        attached (``comm_init``) the call is collective and returns the totals over all ranks."""
        if not self.comm_size:
            self.sync()
        nlib, Ln, A = len(self.libraries), self.length, self.around
        mis = np.zeros((nlib, 2, 2, Ln, L.N_MIS_COLS), np.uint64)
        comp = np.zeros((nlib, 2, 2, Ln + A, 4), np.uint64)
        lgd = np.zeros((nlib, 2, 2, self.lgd_max), np.uint64)
        # (with a communicator the list holds every rank's entries: each rank is bounded by lgd_over_cap)
        cap = max(1, self.lgd_over_cap) * max(1, self.comm_size)
        over = np.zeros((cap, 4), np.int64)
        n_over = ctypes.c_int64(0)
        n_kept = ctypes.c_int64(0)
        rc = self._lib.mdx_finish(self._ctx, _ptr(mis), _ptr(comp), _ptr(lgd), _ptr(over),
                                  ctypes.c_int64(cap), ctypes.byref(n_over), ctypes.byref(n_kept))
        if rc == L.MDX_ERR_BAD_READ:
            self.sync()     # raises BadReadError with the record's index
        self._check(rc)
        return TableSet(self.libraries, Ln, A, mis, comp, lgd, over[:n_over.value].copy(),
                        n_kept.value)

    def lgd_overflow_only(self):
        """Out-of-range fragment-length records of this context (used after an all-reduce)."""
        self.sync()
        cap = max(1, self.lgd_over_cap)
        over = np.zeros((cap, 4), np.int64)
        n_over = ctypes.c_int64(0)
        self._check(self._lib.mdx_finish(self._ctx, None, None, None, _ptr(over), ctypes.c_int64(cap),
                                         ctypes.byref(n_over), None))
        return over[:min(cap, n_over.value)].copy()

    def genome_composition(self, n_contig):
        """Per-contig A, C, G, T counts of the resident reference (seqtk.comp of the reference)."""
        counts = np.zeros((n_contig, 4), np.uint64)
        self._check(self._lib.mdx_genome_composition(self._ctx, _ptr(counts)))
        return counts

    def set_rescale_model(self, model):
        """``model``: mapdamage_amd.rescale.RescaleModel."""
        lut = np.ascontiguousarray(model.lut, dtype=np.uint8)
        term = np.ascontiguousarray(model.term, dtype=np.float64)
        self._check(self._lib.mdx_rescale_set_model(self._ctx, _ptr(lut), _ptr(term),
                                                    ctypes.c_int32(model.len5p), ctypes.c_int32(model.len3p)))

    def rescale(self, batch: ReadBatch):
        """Rescaled qualities, raw MR sums (NaN = record unchanged) and routing status per record
        (mapdamage/rescale.py:285-365).  ``batch`` needs ``qual``, ``mtid`` and ``mpos``."""
        if batch.qual is None or batch.mtid is None or batch.mpos is None:
            raise ValueError("rescaling needs the qual, mtid and mpos columns")
        hb = _host_batch(batch)
        mtid = np.ascontiguousarray(batch.mtid, dtype=np.int32)
        mpos = np.ascontiguousarray(batch.mpos, dtype=np.int32)
        qual_out = np.zeros_like(batch.qual)
        mr = np.zeros(batch.n, np.float64)
        status = np.zeros(batch.n, np.uint8)
        rc = self._lib.mdx_rescale_host(self._ctx, ctypes.byref(hb), _ptr(mtid), _ptr(mpos), _ptr(qual_out),
                                        _ptr(mr), _ptr(status))
        if rc == L.MDX_ERR_BAD_READ:
            self.sync()     # raises BadReadError with the record's index within the batch
        self._check(rc)
        return qual_out, mr, status

    def rescale_device(self, dbatch, d_mtid, d_mpos, d_qual_out, d_mr, d_status, with_tables=False):
        """Rescale a resident batch (``DeviceBatch`` with qualities; the other arguments are device pointers as
        integers); ``with_tables``: count the batch into the tables in the same call (BASELINE configs[4])."""
        fn = self._lib.mdx_tabulate_rescale_device if with_tables else self._lib.mdx_rescale_device
        self._set_record_base(0)
        self._check(fn(self._ctx, ctypes.byref(dbatch.dev), ctypes.c_void_p(d_mtid), ctypes.c_void_p(d_mpos),
                       ctypes.c_void_p(d_qual_out), ctypes.c_void_p(d_mr), ctypes.c_void_p(d_status)))

    def rescale_patches(self, dbatch, d_mtid, d_mpos, d_patch, patch_cap, n_parts, d_n_patch, d_mr, d_status, with_tables=False):
        """``rescale_device`` with the rescaled bytes as a list (include/mdx.h ``mdx_*_patches_device``) in ``n_parts`` parts
        (a power of two): ``d_patch`` a device buffer of ``n_parts * patch_cap`` uint64 entries (byte index | new Phred <<
        32), ``d_n_patch`` ``n_parts`` device uint64 that receive the entries of each part."""
        fn = self._lib.mdx_tabulate_rescale_patches_device if with_tables else self._lib.mdx_rescale_patches_device
        fn.restype = ctypes.c_int
        dev = dbatch.dev if isinstance(dbatch, DeviceBatch) else dbatch
        self._set_record_base(0)
        self._check(fn(self._ctx, ctypes.byref(dev), ctypes.c_void_p(d_mtid), ctypes.c_void_p(d_mpos), ctypes.c_void_p(d_patch),
                       ctypes.c_int64(patch_cap), ctypes.c_int32(n_parts), ctypes.c_void_p(d_n_patch), ctypes.c_void_p(d_mr),
                       ctypes.c_void_p(d_status)))

    def rescale_expand(self, dbatch, d_patch, patch_cap, n_parts, d_n_patch, d_qual_out):
        """The batch's quality column with a patch list applied, into ``d_qual_out`` (device pointers)."""
        dev = dbatch.dev if isinstance(dbatch, DeviceBatch) else dbatch
        self._lib.mdx_rescale_expand_device.restype = ctypes.c_int
        self._check(self._lib.mdx_rescale_expand_device(self._ctx, ctypes.byref(dev), ctypes.c_void_p(d_patch), ctypes.c_int64(patch_cap),
                                                        ctypes.c_int32(n_parts), ctypes.c_void_p(d_n_patch), ctypes.c_void_p(d_qual_out)))

    def fused_launches(self):
        """Calls of rescale_device(with_tables=True) so far that ran as one fused launch."""
        return int(self._lib.mdx_fused_launches(self._ctx))

    def packed_launches(self):
        """Kernel launches so far that ran as the packed kernel (4-bit SEQ column and reference)."""
        return int(self._lib.mdx_packed_launches(self._ctx))

    def libsorts(self):
        """Calls so far that bucketed their batch by library inside the launch (several libraries, a batch that did not
        bring the sorted columns: ``upload`` does)."""
        return int(self._lib.mdx_libsorts(self._ctx))

    def rescale_timing_read(self):
        n = ctypes.c_int64(0)
        ms = ctypes.c_double(0)
        self._check(self._lib.mdx_rescale_timing_read(self._ctx, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    def rescale_summary(self):
        """Integer content of the reference's ``subs`` dictionary (rescale.py:82-143) accumulated since
        set_rescale_model; mapdamage_amd.rescale.RescaleSummary turns it into the log lines."""
        n = int(self._lib.mdx_rescale_summary_words(self._ctx))
        words = np.zeros(n, np.uint64)
        self._check(self._lib.mdx_rescale_summary(self._ctx, _ptr(words)))
        return words

    def reset(self):
        self._check(self._lib.mdx_reset(self._ctx))

    # ------------------------------------------------------------------ timing
    def timing(self, enable=True):
        self._check(self._lib.mdx_timing_enable(self._ctx, 1 if enable else 0))

    def timing_read(self):
        n = ctypes.c_int64(0)
        ms = ctypes.c_double(0)
        self._check(self._lib.mdx_timing_read(self._ctx, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value
