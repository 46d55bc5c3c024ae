"""Seeded synthetic genomes and alignment batches.

The reference ships no test BAM and publishes no benchmark input, so every workload
is generated here: a genome with an ``N`` run and a soft-masked (lower-case) stretch,
and read batches with the post-mortem damage model of the survey (C>T at 5' ends,
G>A at 3' ends, decaying geometrically), emitted directly as SoA ``ReadBatch``
columns so that kernel benchmarks do not depend on a BAM existing on disk.

Two generators:

* ``make_reads`` — numpy-vectorised; pure-``M`` reads plus configurable fractions of
  soft/hard clips, one insertion/deletion/``N`` skip, paired flags, several libraries.
* ``make_edge_reads`` — a small hand-enumerated set that hits every quirk listed in
  SURVEY.md Appendix A/D (contig edges, ``N`` ops, leading insertions, clipped-only
  reads, IUPAC symbols, filtered flags ...).
"""

import numpy as np

from . import layout as L
from .batch import ReadBatch, Reference, batch_from_records, concat_batches

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_genome(seed=20240108, sizes=(("chr1", 8_000_000), ("chr2", 2_000_000), ("chrS", 500)),
                n_run=1000, lower_run=10_000):
    """Genome per SURVEY.md §8d: i.i.d. bases P(A,C,G,T)=(.3,.2,.2,.3), one ``N`` run
    and one lower-case stretch per large contig."""
    rng = np.random.default_rng(seed)
    names, seqs = [], []
    for name, size in sizes:
        codes = rng.choice(4, size=size, p=[0.3, 0.2, 0.2, 0.3]).astype(np.uint8)
        s = _ACGT[codes]
        if size >= 4 * (n_run + lower_run) and n_run:
            a = size // 3
            s[a:a + n_run] = ord("N")
            b = (2 * size) // 3
            s[b:b + lower_run] |= 0x20  # lower-case
        names.append(name)
        seqs.append(s.tobytes())
    return Reference(names, seqs)


def small_genome(seed=7):
    """A genome small enough for per-read Python oracles (three contigs, 6.5 kb)."""
    return make_genome(seed, sizes=(("c1", 4000), ("c2", 2000), ("cS", 500)), n_run=40,
                       lower_run=200)


def _damage_probs(maxlen):
    i = np.arange(maxlen, dtype=np.float64)
    return (0.30 * 0.7 ** i + 0.01).astype(np.float32)


def make_reads(ref, n, seed, read_len=100, len_range=None, nlib=1, paired=False,
               frac_reverse=0.5, frac_softclip=0.0, frac_ins=0.0, frac_del=0.0,
               frac_skip=0.0, frac_hardclip=0.0, frac_filtered=0.0, frac_n_base=0.0,
               with_qual=False, damage=True, contigs=None, chunk=500_000, sort=False, clip_max=10):
    """Vectorised read generator.  Returns a ``ReadBatch``.

    ``len_range=(lo, hi)`` draws SEQ lengths uniformly (config 4); otherwise all reads
    have ``read_len`` bases.  Start positions are uniform over the chosen contigs and at
    least ``12`` bases away from either contig end (edge cases live in ``make_edge_reads``).
    """
    rng = np.random.default_rng(seed)
    bases, offs = ref.concat()
    upper = np.concatenate([bases & np.uint8(0xDF), np.full(1024, ord("N"), np.uint8)])
    lens = np.asarray(ref.lengths, dtype=np.int64)
    if contigs is None:
        contigs = [i for i, ln in enumerate(lens) if ln >= 2000]
    contigs = np.asarray(contigs, dtype=np.int64)
    weights = lens[contigs] / lens[contigs].sum()
    out = []
    done = 0
    while done < n:
        m = min(chunk, n - done)
        out.append(_make_chunk(rng, m, upper, offs, lens, contigs, weights, read_len, len_range,
                               nlib, paired, frac_reverse, frac_softclip, frac_ins, frac_del,
                               frac_skip, frac_hardclip, frac_filtered, frac_n_base, with_qual,
                               damage, clip_max))
        done += m
    batch = out[0] if len(out) == 1 else concat_batches(out)
    if sort:
        order = np.lexsort((batch.pos, batch.tid))
        batch = _permute_fixed(batch, order)
    return batch


def _permute_fixed(batch, order):
    """Reorder records (vectorised; handles ragged columns through a gather index)."""
    def gather(off, data):
        ln = (off[1:] - off[:-1]).astype(np.int64)[order]
        new_off = np.zeros(len(order) + 1, np.int64)
        np.cumsum(ln, out=new_off[1:])
        start = off[:-1].astype(np.int64)[order]
        idx = np.repeat(start - new_off[:-1], ln) + np.arange(new_off[-1])
        return new_off.astype(np.uint32), data[idx]
    coff, cig = gather(batch.cigar_off, batch.cigar)
    soff, seq = gather(batch.seq_off, batch.seq)
    qual = None
    if batch.qual is not None:
        _, qual = gather(batch.seq_off, batch.qual)
    return ReadBatch(batch.flag[order], batch.lib[order], batch.tid[order], batch.pos[order],
                     batch.tlen[order], coff, cig, soff, seq, qual).validate()


def _make_chunk(rng, n, upper, offs, lens, contigs, weights, read_len, len_range, nlib, paired,
                frac_reverse, frac_softclip, frac_ins, frac_del, frac_skip, frac_hardclip,
                frac_filtered, frac_n_base, with_qual, damage, clip_max=10):
    from numpy.lib.stride_tricks import sliding_window_view
    if len_range is None:
        qlen = np.full(n, read_len, dtype=np.int64)
        maxlen = read_len
    else:
        qlen = rng.integers(len_range[0], len_range[1] + 1, size=n).astype(np.int64)
        maxlen = int(len_range[1])

    # clips
    u = rng.random(n)
    has_clip = u < frac_softclip
    side = rng.integers(0, 3, size=n)  # 0 left, 1 right, 2 both
    a = np.where(has_clip & (side != 1), rng.integers(1, clip_max + 1, size=n), 0).astype(np.int64)
    b = np.where(has_clip & (side != 0), rng.integers(1, clip_max + 1, size=n), 0).astype(np.int64)
    over = (a + b) > (qlen - 20)  # keep at least 20 aligned bases
    a[over] = 0
    b[over] = 0
    hard = rng.random(n) < frac_hardclip
    hl = np.where(hard, rng.integers(1, 6, size=n), 0).astype(np.int64)

    # one mid operation: 0 none, 1 I, 2 D, 3 N
    u = rng.random(n)
    t = np.zeros(n, dtype=np.int64)
    t[u < frac_ins] = 1
    t[(u >= frac_ins) & (u < frac_ins + frac_del)] = 2
    t[(u >= frac_ins + frac_del) & (u < frac_ins + frac_del + frac_skip)] = 3
    k = np.where(t == 3, rng.integers(50, 501, size=n), rng.integers(1, 4, size=n))
    k = np.where(t == 0, 0, k).astype(np.int64)
    aligned = qlen - a - b                      # query bases in M/I ops
    ins_k = np.where(t == 1, k, 0)
    mlen = aligned - ins_k                      # bases in M ops
    split = (5 + (rng.random(n) * np.maximum(mlen - 10, 1)).astype(np.int64))
    m1 = np.where(t == 0, mlen, np.minimum(split, mlen - 5))
    m2 = mlen - m1
    skip = np.where((t == 2) | (t == 3), k, 0)
    span = mlen + skip                          # reference span

    # placement
    ci = rng.choice(len(contigs), size=n, p=weights)
    tid = contigs[ci]
    clen = lens[tid]
    margin = 12
    pos = margin + (rng.random(n) * (clen - span - 2 * margin)).astype(np.int64)
    gstart = offs[tid] + pos

    # SEQ matrix: column c of a plain read is reference base gstart - a + c
    win = sliding_window_view(upper, maxlen)
    seqm = win[gstart - a].copy()
    cplx = np.nonzero((a > 0) | (b > 0) | (t > 0))[0]
    if cplx.size:
        c = np.arange(maxlen, dtype=np.int64)[None, :]
        a_, m1_, ins_, q_, b_ = (x[cplx, None] for x in (a, m1, ins_k, qlen, b))
        sub = seqm[cplx]
        # bases after the mid operation come from a window shifted by (skip - inserted)
        sub2 = win[gstart[cplx] - a[cplx] + skip[cplx] - ins_k[cplx]]
        sub = np.where(c >= a_ + m1_ + ins_, sub2, sub)
        rnd_cols = (c < a_) | ((c >= a_ + m1_) & (c < a_ + m1_ + ins_)) | (c >= q_ - b_)
        rnd = _ACGT[rng.integers(0, 4, size=sub.shape, dtype=np.uint8)]
        seqm[cplx] = np.where(rnd_cols, rnd, sub)
    isn = np.nonzero(seqm == ord("N"))
    if isn[0].size:
        seqm[isn] = _ACGT[rng.integers(0, 4, size=isn[0].size, dtype=np.uint8)]

    if damage:
        p = _damage_probs(maxlen + 1)
        r = rng.random((n, maxlen), dtype=np.float32)
        ct = (r < p[None, :maxlen]) & (seqm == ord("C"))      # distance from the left end
        r = rng.random((n, maxlen), dtype=np.float32)
        if len_range is None:
            ga = r < p[None, maxlen - 1::-1][:, :maxlen]       # distance from the right end
        else:
            d3 = np.clip(qlen[:, None] - 1 - np.arange(maxlen)[None, :], 0, maxlen)
            ga = r < p[d3]
        ga &= seqm == ord("G")
        seqm[ct] = ord("T")
        seqm[ga] = ord("A")
        r = rng.random((n, maxlen), dtype=np.float32)
        bg = np.nonzero(r < 0.001)
        seqm[bg] = _ACGT[rng.integers(0, 4, size=bg[0].size, dtype=np.uint8)]
    if frac_n_base > 0:
        rows = rng.random(n) < frac_n_base
        r = rng.random((n, maxlen), dtype=np.float32)
        seqm[rows[:, None] & (r < 0.05)] = ord("N")

    if len_range is None:
        seq = seqm.reshape(-1)
    else:
        seq = seqm[np.arange(maxlen)[None, :] < qlen[:, None]]
    seq_off = np.zeros(n + 1, np.int64)
    np.cumsum(qlen, out=seq_off[1:])
    qual = None
    if with_qual:
        qual = rng.integers(2, 42, size=seq.shape[0], dtype=np.uint8)

    # CIGAR: [H] [S] M1 [X] [M2] [S] [H]
    if cplx.size or hard.any():
        mid_op = np.array([0, L.OP_I, L.OP_D, L.OP_N], dtype=np.int64)[t]
        ops = np.stack([
            (hl << 4) | L.OP_H, (a << 4) | L.OP_S, (m1 << 4) | L.OP_M, (k << 4) | mid_op,
            (m2 << 4) | L.OP_M, (b << 4) | L.OP_S], axis=1)
        present = np.stack([hl > 0, a > 0, m1 > 0, k > 0, m2 > 0, b > 0], axis=1)
        cigar = ops[present].astype(np.uint32)
        cigar_off = np.zeros(n + 1, np.int64)
        np.cumsum(present.sum(axis=1), out=cigar_off[1:])
    else:
        cigar = ((m1 << 4) | L.OP_M).astype(np.uint32)
        cigar_off = np.arange(n + 1, dtype=np.int64)

    flag = np.zeros(n, dtype=np.int64)
    rev = rng.random(n) < frac_reverse
    flag |= np.where(rev, L.FLAG_REVERSE, 0)
    tlen = np.zeros(n, dtype=np.int64)
    if paired:
        flag |= L.FLAG_PAIRED
        flag |= np.where(rng.random(n) < 0.95, L.FLAG_PROPER, 0)
        flag |= np.where(rng.random(n) < 0.5, L.FLAG_READ1, 0x80)
        flag |= np.where(~rev, 0x20, 0)
        tl = np.maximum(100, rng.normal(180, 30, size=n)).astype(np.int64)
        tlen = np.where(rev, -tl, tl)
    if frac_filtered > 0:
        u = rng.random(n)
        bad = u < frac_filtered
        which = np.array([L.FLAG_UNMAPPED, L.FLAG_SECONDARY, L.FLAG_QCFAIL, L.FLAG_DUP,
                          L.FLAG_SUPPLEMENTARY], dtype=np.int64)[rng.integers(0, 5, size=n)]
        flag |= np.where(bad, which, 0)
    lib = rng.integers(0, nlib, size=n) if nlib > 1 else np.zeros(n, np.int64)

    return ReadBatch(flag.astype(np.uint16), lib.astype(np.uint16), tid.astype(np.int32),
                     pos.astype(np.int32), tlen.astype(np.int32), cigar_off.astype(np.uint32),
                     cigar, seq_off.astype(np.uint32), np.ascontiguousarray(seq), qual).validate()


def make_edge_reads(ref, with_qual=True, nlib=2):
    """Hand-enumerated records exercising every quirk of SURVEY.md Appendix A / D.

    ``ref`` must have a short last contig (``small_genome``'s ``cS``)."""
    rng = np.random.default_rng(99)
    seqs = [s.upper() for s in ref.seqs]
    lens = ref.lengths
    recs = []

    def add(tid, pos, cigar, flag=0, seq=None, tlen=0, lib=0, qual="rand", mutate=()):
        ops = _parse_cigar(cigar)
        qn = sum(ln for op, ln in ops if op in (L.OP_M, L.OP_I, L.OP_S, L.OP_EQ, L.OP_X))
        if seq is None:
            # copy the reference through M/=/X, random elsewhere
            out = []
            rp = pos
            for op, ln in ops:
                if op in (L.OP_M, L.OP_EQ, L.OP_X):
                    chunk = seqs[tid][rp:rp + ln].decode()
                    chunk += "A" * (ln - len(chunk))
                    chunk = "".join(ch if ch in "ACGT" else "ACGT"[rng.integers(0, 4)]
                                    for ch in chunk)
                    out.append(chunk)
                    rp += ln
                elif op in (L.OP_I, L.OP_S):
                    out.append("".join("ACGT"[i] for i in rng.integers(0, 4, size=ln)))
                elif op in (L.OP_D, L.OP_N):
                    rp += ln
            seq = "".join(out)
            seq = list(seq)
            for i, ch in mutate:
                if i < len(seq):
                    seq[i] = ch
            seq = "".join(seq)
        assert len(seq) == qn, (cigar, len(seq), qn)
        q = None
        if with_qual and qual is not None:
            q = [int(x) for x in rng.integers(2, 42, size=len(seq))] if qual == "rand" else qual
        recs.append(dict(flag=flag, lib=lib % nlib, tid=tid, pos=pos, tlen=tlen, cigar=ops,
                         seq=seq, qual=q))

    R = L.FLAG_REVERSE
    last = len(lens) - 1
    n_last = lens[last]
    for strand in (0, R):
        # plain matches, both strands, different lengths (shorter than L, longer than 2L)
        add(0, 100, "30M", strand)
        add(0, 200, "75M", strand, lib=1)
        add(0, 300, "150M", strand)
        add(0, 350, "1M", strand)
        add(0, 351, "2M", strand, lib=1)
        # mismatches at both ends (all 12 substitutions appear through mutate cycling)
        add(0, 400, "40M", strand, mutate=[(0, "T"), (1, "A"), (2, "C"), (3, "G"), (36, "T"),
                                           (37, "A"), (38, "C"), (39, "G")])
        # soft clips: left, right, both, longer than L is impossible at L=70 but fine at L=8
        add(0, 500, "5S35M", strand)
        add(0, 520, "35M7S", strand, lib=1)
        add(0, 540, "3S30M12S", strand)
        # hard clips alone and with soft clips (leading H then S keeps column 0)
        add(0, 600, "2H36M", strand)
        add(0, 620, "2H4S30M3S1H", strand, lib=1)
        # insertions / deletions incl. leading and trailing insertion
        add(0, 700, "10M2I20M", strand)
        add(0, 720, "10M3D20M", strand, lib=1)
        add(0, 740, "2I30M", strand)
        add(0, 760, "30M2I", strand)
        add(0, 780, "5M1I5M1D5M2I5M2D10M", strand)
        add(0, 800, "3S5M1I20M2D6M4S", strand, lib=1)
        # N skip: columns after the skip are mis-aligned by design (Appendix A.2)
        add(0, 900, "12M60N15M", strand)
        add(0, 1000, "4M5N4M", strand, lib=1)
        add(0, 1020, "6M2I4M30N8M1D6M", strand)
        # padding op and '=' / 'X' ops
        add(0, 1100, "10M2P10M", strand)
        add(0, 1120, "10=1X10=", strand, lib=1)
        # reads with N / IUPAC symbols and '=' in SEQ
        add(0, 1200, "20M", strand, mutate=[(1, "N"), (4, "R"), (18, "N"), (19, "=")])
        # contig start / end (short or empty flanks)
        add(0, 0, "25M", strand)
        add(0, 3, "25M", strand, lib=1)
        add(last, n_last - 25, "25M", strand)
        add(last, n_last - 28, "25M", strand)
        # (an alignment running past the contig end makes the reference raise ValueError from
        # FastaFile.fetch(start > end) in align.py:33; covered by the error tests, not here)
        # no reference-consuming op at all (htslib: aend = pos + 1)
        add(0, 1300, "12S", strand)
        add(0, 1310, "6I", strand)
        add(0, 1320, "3S4I2S", strand, lib=1)
        # reads over the N run and the lower-case stretch
        add(0, lens[0] // 3 - 10, "40M", strand)
        add(0, (2 * lens[0]) // 3 - 10, "40M", strand, lib=1)
        # paired records
        P = L.FLAG_PAIRED
        add(0, 1400, "30M", strand | P | L.FLAG_PROPER | L.FLAG_READ1, tlen=-180 if strand else 180)
        add(0, 1430, "30M", strand | P | L.FLAG_PROPER | 0x80, tlen=175)
        add(0, 1460, "30M", strand | P | L.FLAG_READ1, tlen=0, lib=1)
        add(0, 1490, "30M", strand | P | L.FLAG_PROPER | L.FLAG_READ1, tlen=0)
        add(0, 1500, "30M", strand | P | L.FLAG_PROPER | L.FLAG_READ1, tlen=70000, lib=1)
        # reads without qualities
        add(0, 1600, "30M", strand, qual=None)
        add(0, 1640, "10M2I10M2D8M", strand, qual=None, lib=1)
        # filtered records (must not be counted)
        for bad in (L.FLAG_UNMAPPED, L.FLAG_SECONDARY, L.FLAG_QCFAIL, L.FLAG_DUP,
                    L.FLAG_SUPPLEMENTARY):
            add(0, 1700, "30M", strand | bad)
        # zero-length SEQ is legal in BAM ("*"); with an M op pysam yields query '' -> no pairs
    return batch_from_records(recs, with_qual=with_qual)


def _parse_cigar(text):
    ops, num = [], ""
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            ops.append((L.CIGAR_CHARS.index(ch), int(num)))
            num = ""
    return ops


def config1_batch(ref=None, seed=1, n=1000):
    """BASELINE config[0]/survey config 1: ~1k mixed reads, 2 libraries, all op kinds."""
    ref = ref or small_genome()
    mixed = make_reads(ref, n, seed, len_range=(30, 120), nlib=2, frac_softclip=0.10,
                       frac_ins=0.05, frac_del=0.05, frac_skip=0.01, frac_hardclip=0.01,
                       frac_filtered=0.05, frac_n_base=0.01, with_qual=True)
    edge = make_edge_reads(ref, with_qual=True, nlib=2)
    return ref, concat_batches([mixed, edge])


def config2_batch(ref, n=5_000_000, seed=2):
    """Survey config 2: SE 100 bp ``100M`` reads, one library, damage model."""
    return make_reads(ref, n, seed, read_len=100, contigs=[0, 1])


def config3_batch(ref, n, seed=3, with_qual=False):
    return make_reads(ref, n, seed, read_len=100, paired=True, frac_softclip=0.10, frac_ins=0.04,
                      frac_del=0.04, frac_skip=0.002, frac_hardclip=0.001, contigs=[0, 1],
                      with_qual=with_qual)


_PAR_REF = None


def _par_worker(job):
    maker, n, seed = job
    if isinstance(maker, dict):      # keyword arguments of make_reads
        return make_reads(_PAR_REF, n, seed, **maker)
    return globals()[maker](_PAR_REF, n, seed=seed)


def parallel_batch(maker, ref, n, seed, workers=None, shard=500_000):
    """``maker`` ("config2_batch" | "config3_batch" | "config4_batch", or a dict of ``make_reads`` keyword
    arguments) over ``n`` records as independent shards
    of ``shard`` records, shard k seeded ``[seed, k]``, generated on a pool of forked worker processes and
    concatenated in shard order — the full-size workloads (50 M records) in tens of seconds instead of minutes.
    Deterministic in (maker, n, seed, shard); NOT the same records as ``maker(ref, n, seed)``.
    Call it before the process touches the GPU (the workers are forked)."""
    import multiprocessing as mp
    import os
    global _PAR_REF
    jobs = [(maker, min(shard, n - lo), [seed, k]) for k, lo in enumerate(range(0, n, shard))]
    from .sam import usable_cpus
    avail = usable_cpus()
    workers = max(1, min(workers or avail, len(jobs)))
    _PAR_REF = ref
    try:
        if workers == 1:
            parts = [_par_worker(j) for j in jobs]
        else:
            with mp.get_context("fork").Pool(workers) as pool:
                parts = pool.map(_par_worker, jobs, chunksize=1)
    finally:
        _PAR_REF = None
    return parts[0] if len(parts) == 1 else concat_batches(parts)


def config4_batch(ref, n, seed=4):
    return make_reads(ref, n, seed, len_range=(35, 150), paired=True, frac_softclip=0.10,
                      frac_ins=0.04, frac_del=0.04, frac_skip=0.002, frac_hardclip=0.001,
                      contigs=[0, 1])
