"""A BAM file that mapdamage_amd's own writer (sam.write_bam) did NOT produce: every byte assembled here from the layout
of the SAM/BAM specification (SAMv1 section 4: BGZF framing 4.1, header and record layout 4.2, the CG tag of 4.2.2), so
that the decoders — the host one (csrc/mdx_bamio.cpp), its device counterpart (csrc/mdx_gbam.hip) and the Python
cross-check (sam.read_bam) — are held against construction-time truth instead of against a writer that shares their
author's reading of the format.  What pysam hands the reference for such records is SURVEY Appendix C
(/root/reference/mapdamage/reader.py:63-81,99-118 for the read groups).

In the file: a header with several read groups (two libraries of one sample, the same library name under another
sample, extra tags on the @RG lines, @PG / @CO lines), header and records as ONE stream cut into BGZF blocks of uneven
sizes anywhere (the header shares a block with records; records straddle blocks), `=` / `X` / `I` / `D` / `N` / `P` /
`S` / `H` operations, aux fields of every type (`A c C s S i I f Z H` and `B` arrays of every subtype) in front of and
behind `RG:Z`, a record without RG, records without qualities (0xFF), a secondary record without SEQ (l_seq = 0), an
unmapped record, a record of 33 000 bases whose 66 000 CIGAR operations live in `CG:B,I` behind the `<l_seq>S<n>N`
placeholder, MAPQ 255, and NO end-of-file marker block.

Writes tests/golden/foreign.bam (everything), foreign_nocg.bam (without the long record: the device path takes it),
foreign_nolb.bam (an @RG line without LB added: the reference's BAMError) and foreign_bam.npz (genome + truth).
    python tools/make_foreign_bam.py
"""
import pathlib
import struct
import zlib

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
OUT = ROOT / "tests" / "golden"
CONTIGS = (("chrA", 70_000), ("chrB", 1_500))
SEQ_CODES = "=ACMGRSVTWYHKDBN"
OPS = "MIDNSHP=X"


def reg2bin(beg, end):
    """SAMv1 section 5.3."""
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def bgzf_block(payload):
    """One BGZF block (SAMv1 4.1): a gzip member with the `BC` extra subfield holding the block's size less one."""
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    data = comp.compress(payload) + comp.flush()
    bsize = 12 + 6 + len(data) + 8 - 1
    return (b"\x1f\x8b\x08\x04" + struct.pack("<IBBH", 0, 0, 0xFF, 6) + b"BC" + struct.pack("<HH", 2, bsize) + data +
            struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))


def aux(tag, typ, value, sub=None):
    t = tag.encode() + typ.encode()
    if typ == "A":
        return t + value.encode()
    if typ in "cCsSiIf":
        return t + struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[typ], value)
    if typ in "ZH":
        return t + value.encode() + b"\x00"
    assert typ == "B"
    fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
    return t + sub.encode() + struct.pack("<i", len(value)) + struct.pack("<%d%s" % (len(value), fmt), *value)


def record(name, flag, tid, pos, mapq, cigar, seq, qual, mtid, mpos, tlen, tags):
    """cigar: [(op letter, length)]; seq: str or "" (absent); qual: bytes, None = 0xFF..; tags: encoded aux bytes."""
    l_seq = len(seq)
    ops = [(n << 4) | OPS.index(op) for op, n in cigar]
    ref_len = sum(n for op, n in cigar if op in "MDN=X")
    extra = b""
    if len(ops) > 65535:
        # SAMv1 4.2.2: the operations go to CG:B,I, the field keeps <l_seq>S<reference length>N
        extra = aux("CG", "B", ops, "I")
        ops = [(l_seq << 4) | 4, (ref_len << 4) | 3]
    packed = bytearray((l_seq + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i >> 1] |= SEQ_CODES.index(ch) << (0 if i & 1 else 4)
    q = bytes([0xFF] * l_seq) if qual is None else bytes(qual)
    assert len(q) == l_seq
    end = pos + (ref_len if ref_len else 1)
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(name) + 1, mapq, reg2bin(max(pos, 0), max(end, 1)) if tid >= 0 else 4680,
                       len(ops), flag, l_seq, mtid, mpos, tlen)
    body += name.encode() + b"\x00" + struct.pack("<%dI" % len(ops), *ops) + bytes(packed) + q + tags + extra
    return struct.pack("<i", len(body)) + body


def build(with_long=True, with_nolb=False, seed=20240917):
    rng = np.random.default_rng(seed)
    genome = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n, p=(0.3, 0.2, 0.2, 0.3)) for _, n in CONTIGS]
    genome[0][5000:5040] = ord("N")
    genome[0][6000:6200] |= 0x20           # a soft-masked stretch
    header = ["@HD\tVN:1.6\tSO:unsorted"] + ["@SQ\tSN:%s\tLN:%d" % c for c in CONTIGS] + [
        "@RG\tID:lane1\tPL:ILLUMINA\tSM:sampleX\tLB:libA\tDS:first lane",
        "@RG\tID:lane.2\tSM:sampleX\tLB:libB",
        "@RG\tID:L3\tLB:libA\tSM:sampleY"]
    if with_nolb:
        header.append("@RG\tID:nolb\tSM:sampleZ\tPU:unit7")
    header += ["@PG\tID:handmade\tPN:make_foreign_bam\tCL:python tools/make_foreign_bam.py", "@CO\tassembled byte by byte"]
    text = ("\n".join(header) + "\n").encode()
    stream = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(CONTIGS)))
    for name, ln in CONTIGS:
        stream += struct.pack("<i", len(name) + 1) + name.encode() + b"\x00" + struct.pack("<i", ln)

    truth = []          # (name, flag, tid, pos, tlen, mtid, mpos, ops, seq, qual or None, rg or None)

    def query(tid, pos, cigar, damage=0.05):
        """Read bases for a CIGAR at (tid, pos): the reference under M / = operations (X: a mismatch), random bases for
        I and S; a few bases changed."""
        out, p = [], pos
        g = genome[tid]
        for op, n in cigar:
            if op in "M=":
                seg = (g[p:p + n] & 0xDF).copy()
                if op == "M":
                    hit = rng.random(n) < damage
                    seg[hit] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(hit.sum()))
                out.append(seg); p += n
            elif op == "X":
                seg = (g[p:p + n] & 0xDF).copy()
                seg = np.where(seg == ord("A"), ord("C"), ord("A")).astype(np.uint8)
                out.append(seg); p += n
            elif op in "IS":
                out.append(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n))
            elif op in "DN":
                p += n
        s = np.concatenate(out) if out else np.zeros(0, np.uint8)
        return bytes(s.astype(np.uint8)).decode()

    def add(name, flag, tid, pos, cigar, rg, tags_front=b"", tags_back=b"", qual="rand", seq=None, mtid=-1, mpos=-1, tlen=0, mapq=37):
        s = query(tid, pos, cigar) if seq is None else seq
        q = None if qual is None else bytes(rng.integers(2, 42, size=len(s)).astype(np.uint8))
        tags = tags_front + (aux("RG", "Z", rg) if rg is not None else b"") + tags_back
        stream.extend(record(name, flag, tid, pos, mapq, cigar, s, q, mtid, mpos, tlen, tags))
        truth.append((name, flag & 0x7FFF, tid, pos, tlen, mtid, mpos, [(OPS.index(op), n) for op, n in cigar], s, q, rg))

    every = (aux("XA", "A", "Q") + aux("Xc", "c", -5) + aux("XC", "C", 250) + aux("Xs", "s", -300) + aux("XS", "S", 60000) +
             aux("Xi", "i", -70000) + aux("XI", "I", 4000000000) + aux("Xf", "f", 2.5) + aux("XZ", "Z", "text with RG:Z:lane1 inside") +
             aux("XH", "H", "1AE301"))
    arrays = (aux("Bc", "B", [-1, 2, -3], "c") + aux("BC", "B", [1, 2, 255], "C") + aux("Bs", "B", [-300, 300], "s") +
              aux("BS", "B", [65535], "S") + aux("Bi", "B", [-1, 70000], "i") + aux("BI", "B", [1, 2, 3, 4], "I") +
              aux("Bf", "B", [0.5, -1.5], "f") + aux("Be", "B", [], "C"))
    add("plain_fwd", 0, 0, 100, [("M", 40)], "lane1")
    add("plain_rev", 16, 0, 300, [("M", 55)], "lane.2", tags_back=aux("NM", "C", 2))
    add("eq_x", 0, 0, 500, [("=", 12), ("X", 1), ("=", 22)], "L3", tags_front=aux("NM", "C", 1) + aux("MD", "Z", "12A22"))
    add("clips", 16, 0, 700, [("H", 5), ("S", 3), ("M", 30), ("S", 7)], "lane1", tags_front=arrays)
    add("ins", 0, 0, 900, [("M", 20), ("I", 2), ("M", 25)], "lane.2", tags_front=every)
    add("del", 16, 0, 1100, [("M", 18), ("D", 3), ("M", 30)], "lane1", tags_front=every, tags_back=arrays)
    add("skip_pad", 0, 0, 1300, [("M", 15), ("N", 200), ("M", 10), ("P", 2), ("I", 1), ("M", 12)], "L3")
    add("no_qual", 0, 0, 1700, [("M", 35)], "lane1", qual=None)
    add("no_qual_rev", 16, 1, 200, [("S", 2), ("M", 33)], "lane.2", qual=None)
    add("secondary_no_seq", 256, 0, 100, [("M", 40)], "lane1", seq="", tags_front=aux("NH", "i", 2))
    add("unmapped", 4, -1, -1, [], "lane1", seq="ACGTNACGT", mapq=0)
    add("no_rg_duplicate", 1024, 0, 2100, [("M", 30)], None)
    add("over_n_run", 0, 0, 4990, [("M", 60)], "L3", mapq=255)
    add("soft_masked", 16, 0, 6010, [("M", 70)], "lane1")
    add("pair_r1", 99, 1, 400, [("M", 50)], "lane.2", mtid=1, mpos=520, tlen=170)
    add("pair_r2", 147, 1, 520, [("M", 50)], "lane.2", mtid=1, mpos=400, tlen=-170)
    add("contig_start", 0, 1, 0, [("M", 25)], "L3")
    add("contig_end", 16, 1, 1500 - 28, [("M", 28)], "L3")
    add("iupac", 0, 0, 2500, [("M", 16)], "lane1", seq="ACGTRYSWKMBDHVN=")
    if with_long:
        add("long_cigar", 0, 0, 2800, [("M", 1), ("D", 1)] * 33_000, "lane1")
    add("after_long", 16, 0, 69_900, [("M", 90)], "lane.2", tags_front=arrays + every)
    add("qcfail", 512, 0, 3000, [("M", 30)], "lane1")
    add("supplementary", 2048, 0, 3100, [("H", 20), ("M", 30)], "L3")
    for k in range(40):
        add("bulk%02d" % k, 16 * (k & 1), k % 2, 50 + 31 * k, [("M", 20 + k)], ("lane1", "lane.2", "L3")[k % 3])

    # the stream cut into BGZF blocks of uneven sizes anywhere; no end-of-file marker
    cuts, sizes, at, k = [], (313, 4001, 65280, 1000, 65280, 17, 65280), 0, 0
    while at < len(stream):
        cuts.append(bytes(stream[at:at + sizes[k % len(sizes)]]))
        at += sizes[k % len(sizes)]
        k += 1
    return b"".join(bgzf_block(c) for c in cuts), truth, genome, header


def truth_arrays(truth):
    cig_off, seq_off, cig, seq, qual = [0], [0], [], [], []
    for (_n, _f, _t, _p, _tl, _mt, _mp, ops, s, q, _rg) in truth:
        cig += [(n << 4) | op for op, n in ops]
        cig_off.append(len(cig))
        seq.append(np.frombuffer(s.encode(), np.uint8))
        qual.append(np.full(len(s), 0xFF, np.uint8) if q is None else np.frombuffer(q, np.uint8))
        seq_off.append(seq_off[-1] + len(s))
    cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0, np.uint8)   # noqa: E731
    return dict(name=np.array([t[0] for t in truth]), flag=np.array([t[1] for t in truth], np.uint16),
                tid=np.array([t[2] for t in truth], np.int32), pos=np.array([t[3] for t in truth], np.int32),
                tlen=np.array([t[4] for t in truth], np.int32), mtid=np.array([t[5] for t in truth], np.int32),
                mpos=np.array([t[6] for t in truth], np.int32), cigar_off=np.array(cig_off, np.uint32),
                cigar=np.array(cig, np.uint32), seq_off=np.array(seq_off, np.uint32), seq=cat(seq), qual=cat(qual),
                rg=np.array(["" if t[10] is None else t[10] for t in truth]), has_rg=np.array([t[10] is not None for t in truth]))


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    out = {}
    for tag, kw in (("full", {}), ("nocg", dict(with_long=False)), ("nolb", dict(with_long=False, with_nolb=True))):
        data, truth, genome, header = build(**kw)
        path = OUT / ("foreign.bam" if tag == "full" else "foreign_%s.bam" % tag)
        path.write_bytes(data)
        if tag != "nolb":
            for k, v in truth_arrays(truth).items():
                out["%s_%s" % (tag, k)] = v
        print(path, len(data), "bytes,", len(truth), "records")
    for (name, _), g in zip(CONTIGS, genome):
        out["genome_" + name] = g
    out["header_text"] = np.array("\n".join(header) + "\n")
    np.savez_compressed(OUT / "foreign_bam.npz", **out)


if __name__ == "__main__":
    main()
