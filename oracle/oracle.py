"""ctypes binding of oracle/libmdx_oracle.so.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product path (mapdamage_amd/) never imports this.
"""

import ctypes
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_LIB = None


class _Params(ctypes.Structure):
    _fields_ = [("length", ctypes.c_int32), ("around", ctypes.c_int32),
                ("minqual", ctypes.c_int32), ("nlib", ctypes.c_int32),
                ("lgd_max", ctypes.c_int32), ("n_contig", ctypes.c_int32)]


def build(force=False):
    so = _HERE / "libmdx_oracle.so"
    src = _HERE / "mdx_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-s", "-C", str(_HERE), "libmdx_oracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(str(build()))
        _LIB.mdx_oracle_tabulate.restype = ctypes.c_int
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class OracleError(RuntimeError):
    def __init__(self, code, read_index):
        super().__init__("oracle error %d at read %d" % (code, read_index))
        self.code = code
        self.read_index = read_index


def tabulate(ref, batch, nlib, length, around, minqual=0, lgd_max=65536):
    """Run the C oracle.  Returns dict(mis, comp, lgd (dense), lgd_over, n_kept) with tables
    indexed by library *id* (not sorted), canonical layout of mapdamage_amd/layout.py."""
    bases, offs = ref.concat()
    P = _Params(length, around, minqual, nlib, lgd_max, len(ref.names))
    mis = np.zeros((nlib, 2, 2, length, 25), np.uint64)
    comp = np.zeros((nlib, 2, 2, length + around, 4), np.uint64)
    lgd = np.zeros((nlib, 2, 2, lgd_max), np.uint64)
    cap = max(1, batch.n)
    over = np.zeros((cap, 4), np.int64)
    n_over = ctypes.c_int64(0)
    n_kept = ctypes.c_int64(0)
    bad = ctypes.c_int64(-1)
    rc = _lib().mdx_oracle_tabulate(
        ctypes.byref(P), _p(bases), _p(offs), ctypes.c_int64(batch.n), _p(batch.flag),
        _p(batch.lib), _p(batch.tid), _p(batch.pos), _p(batch.tlen), _p(batch.cigar_off),
        _p(batch.cigar), _p(batch.seq_off), _p(batch.seq), _p(batch.qual), _p(mis), _p(comp),
        _p(lgd), _p(over), ctypes.c_int64(cap), ctypes.byref(n_over), ctypes.byref(n_kept),
        ctypes.byref(bad))
    if rc != 0:
        raise OracleError(rc, bad.value)
    return dict(mis=mis, comp=comp, lgd=lgd, lgd_over=over[:n_over.value].copy(),
                n_kept=n_kept.value)


def tabulate_parallel(ref, batch, nlib, length, around, minqual=0, lgd_max=65536, threads=None):
    """tabulate() over contiguous slices of the batch on `threads` host threads (ctypes releases the
    GIL around the C call; the C function keeps no state), tables summed.  The all-cores CPU baseline of
    bench.py; same results as tabulate()."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(threads or avail, batch.n or 1))
    if threads == 1:
        return tabulate(ref, batch, nlib, length, around, minqual, lgd_max), 1
    cuts = [batch.n * t // threads for t in range(threads + 1)]
    parts = [batch.slice(cuts[t], cuts[t + 1]) for t in range(threads)]
    with ThreadPoolExecutor(threads) as pool:
        res = list(pool.map(lambda b: tabulate(ref, b, nlib, length, around, minqual, lgd_max), parts))
    out = res[0]
    for r in res[1:]:
        for k in ("mis", "comp", "lgd"):
            out[k] += r[k]
        out["lgd_over"] = np.concatenate([out["lgd_over"], r["lgd_over"]])
        out["n_kept"] += r["n_kept"]
    return out, threads


def rescale(ref, batch, corr, len5p, len3p):
    """C oracle of mapdamage/rescale.py.  corr: float64 [2][1 + len5p + len3p] (see mdx_oracle.c).
    Returns (qual_out u8[n_bases], mr_raw f64[n] (NaN = not rescaled), status u8[n])."""
    bases, offs = ref.concat()
    lib = _lib()
    lib.mdx_oracle_rescale.restype = ctypes.c_int
    corr = np.ascontiguousarray(corr, dtype=np.float64)
    assert corr.shape == (2, 1 + len5p + len3p)
    qual_out = np.zeros_like(batch.qual)
    mr = np.zeros(batch.n, np.float64)
    status = np.zeros(batch.n, np.uint8)
    bad = ctypes.c_int64(-1)
    rc = lib.mdx_oracle_rescale(
        _p(bases), _p(offs), ctypes.c_int32(len(ref.names)), ctypes.c_int64(batch.n), _p(batch.flag),
        _p(batch.tid), _p(batch.pos), _p(batch.cigar_off), _p(batch.cigar), _p(batch.seq_off),
        _p(batch.seq), _p(batch.qual), _p(batch.mtid), _p(batch.mpos), _p(corr), ctypes.c_int32(len5p),
        ctypes.c_int32(len3p), _p(qual_out), _p(mr), _p(status), ctypes.byref(bad))
    if rc != 0:
        raise OracleError(rc, bad.value)
    return qual_out, mr, status


def rescale_with_subs(ref, batch, corr, len5p, len3p):
    """rescale() plus the reference's ``subs`` dictionary (rescale.py:82-143) as
    (counts u64 [4 + 4*2*130], pvals f64 [6]): see record_subs in mdx_oracle.c."""
    bases, offs = ref.concat()
    lib = _lib()
    lib.mdx_oracle_rescale_subs.restype = ctypes.c_int
    corr = np.ascontiguousarray(corr, dtype=np.float64)
    qual_out = np.zeros_like(batch.qual)
    mr = np.zeros(batch.n, np.float64)
    status = np.zeros(batch.n, np.uint8)
    counts = np.zeros(4 + 4 * 2 * 130, np.uint64)
    pvals = np.zeros(6, np.float64)
    bad = ctypes.c_int64(-1)
    rc = lib.mdx_oracle_rescale_subs(
        _p(bases), _p(offs), ctypes.c_int32(len(ref.names)), ctypes.c_int64(batch.n), _p(batch.flag),
        _p(batch.tid), _p(batch.pos), _p(batch.cigar_off), _p(batch.cigar), _p(batch.seq_off),
        _p(batch.seq), _p(batch.qual), _p(batch.mtid), _p(batch.mpos), _p(corr), ctypes.c_int32(len5p),
        ctypes.c_int32(len3p), _p(qual_out), _p(mr), _p(status), ctypes.byref(bad), _p(counts), _p(pvals))
    if rc != 0:
        raise OracleError(rc, bad.value)
    return qual_out, mr, status, counts, pvals


def subs_log_lines(counts, pvals):
    """_qual_summary_subs + _print_subs (rescale.py:146-192) over the oracle's subs."""
    hist = np.asarray(counts[4:]).reshape(4, 2, 130)
    names = ("CT", "TC", "GA", "AG")
    base = dict(zip("ACGT", (int(c) for c in counts[:4])))
    pv = {"CT": (pvals[1], pvals[0]), "TC": (pvals[2], pvals[2]), "GA": (pvals[4], pvals[3]), "AG": (pvals[5], pvals[5])}
    lines = ["Expected substition frequencies before and after rescaling:"]
    for sub in names:
        n = base[sub[0]]
        if n:
            lines.append("    %s>%s    %.4f    %.4f" % (sub[0], sub[1], pv[sub][0] / n, pv[sub][1] / n))
        else:
            lines.append("\t%s\tNA\t\tNA" % (sub,))
    lines.append("Quality metrics before and after scaling:")
    for sub in ("CT", "GA"):
        i = names.index(sub)
        for lv in (0, 10, 20, 30, 40):
            lines.append("    %s-Q%02i% 10i% 10i" % (sub, lv, int(hist[i, 0, lv:].sum()), int(hist[i, 1, lv:].sum())))
    return lines



def rescale_parallel(ref, batch, corr, len5p, len3p, threads=None):
    """rescale() over contiguous slices of the batch on `threads` host threads (the C function keeps no state)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(threads or avail, batch.n or 1))
    cuts = [batch.n * t // threads for t in range(threads + 1)]
    parts = [batch.slice(cuts[t], cuts[t + 1]) for t in range(threads)]
    with ThreadPoolExecutor(threads) as pool:
        res = list(pool.map(lambda b: rescale(ref, b, corr, len5p, len3p), parts))
    return (np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res]),
            np.concatenate([r[2] for r in res]), threads)
