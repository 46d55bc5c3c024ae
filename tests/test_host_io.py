"""Host side of the boundary (CPU only): SAM/BAM/FASTA codecs, read-group / downsampling logic
(mirror of mapdamage/reader.py), the accumulator-class mirror and the emitters."""

import random

import numpy as np
import pytest

from mapdamage_amd import fasta, sam, synth
from mapdamage_amd.reader import BAMReader
from mapdamage_amd.sam import BAMError
from mapdamage_amd.statistics import (DNAComposition, FragmentLengths, MisincorporationRates,
                                      check_table_and_warn_if_dmg_freq_is_low)
from tests.util import GOLDEN, Golden, oracle_tableset

RGS = [{"ID": "rgA", "SM": "Zed", "LB": "libB"}, {"ID": "rgB", "SM": "Alpha", "LB": "libA"},
       {"ID": "rgC", "SM": "Zed", "LB": "libB"}]


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("io")
    ref, batch = synth.config1_batch()
    rg_of = [("rgA", "rgB", "rgC")[int(l) * 2 % 3] if False else ("rgA" if int(l) == 0 else "rgB") for l in batch.lib]
    # a few reads of library 0 use the second read group that names the same (SM, LB)
    rg_of = ["rgC" if (r == "rgA" and i % 7 == 0) else r for i, r in enumerate(rg_of)]
    sam.write_sam(d / "x.sam", batch, ref.names, ref.lengths, RGS, rg_of)
    sam.write_bam(d / "x.bam", batch, ref.names, ref.lengths, RGS, rg_of)
    fasta.write_fasta(d / "ref.fa", ref)
    return d, ref, batch, rg_of


@pytest.mark.parametrize("name", ["x.sam", "x.bam"])
def test_alignment_codecs_roundtrip(files, name):
    d, ref, batch, rg_of = files
    al = sam.read_alignments(d / name)
    for k in ("flag", "tid", "pos", "tlen", "cigar_off", "cigar", "seq_off", "seq", "qual"):
        np.testing.assert_array_equal(getattr(al.batch, k), getattr(batch, k), err_msg=k)
    assert al.rg == rg_of
    assert al.header.references == ref.names and al.header.lengths == ref.lengths


def test_fasta_roundtrip_and_index(files):
    d, ref, _, _ = files
    names, seqs = fasta.read_fasta(d / "ref.fa")
    assert names == ref.names and seqs == ref.seqs
    fai = fasta.read_fasta_index(str(d / "ref.fa.fai"))
    assert fai == dict(zip(ref.names, ref.lengths))
    assert fasta.compare_sequence_dicts(fai, dict(zip(ref.names, ref.lengths)))
    assert not fasta.compare_sequence_dicts(fai, {"c1": 3999})
    assert not fasta.compare_sequence_dicts(fai, {"other": 10})
    reordered = fasta.reference_in_memory(d / "ref.fa", ref.names[::-1])
    assert reordered.seqs == ref.seqs[::-1]
    # (an uncompressed FASTA stays on disk: the library loads it — tests/test_gpu_parity.py holds that loader against this one)
    on_disk = fasta.reference_for_bam(d / "ref.fa", ref.names[::-1])
    assert on_disk.path == str(d / "ref.fa") and on_disk.names == ref.names[::-1] and on_disk.lengths == ref.lengths[::-1]
    with pytest.raises(KeyError):
        fasta.reference_for_bam(d / "ref.fa", ["nope"])
    assert fasta.reference_for_bam(d / "ref.fa", ["nope", ref.names[0]], missing_ok=True).lengths == [0, ref.lengths[0]]


def test_reader_libraries_filter_and_errors(files):
    d, ref, batch, rg_of = files
    r = BAMReader(d / "x.bam")
    assert r.get_libraries() == [("Zed", "libB"), ("Alpha", "libA")]   # rgC merges into the first
    idx = r.kept_indices()
    assert len(idx) == int(((batch.flag & 0xF04) == 0).sum())
    lib = r.library_column(idx)
    np.testing.assert_array_equal(lib, batch.lib[idx])
    merged = BAMReader(d / "x.bam", merge_libraries=True)
    assert merged.get_libraries() == [("*", "*")] and not merged.library_column(idx).any()
    r.handle.rg[int(idx[3])] = None
    with pytest.raises(BAMError, match="has no read-group"):
        r.library_column(idx)
    r.handle.rg[int(idx[3])] = "nope"
    with pytest.raises(BAMError, match="not listed in BAM header"):
        r.library_column(idx)


def test_reader_downsampling_follows_python_rng(files):
    d, ref, batch, _ = files
    kept = np.nonzero((batch.flag & 0xF04) == 0)[0]
    r = BAMReader(d / "x.sam", downsample_to=0.25, downsample_seed=7)
    rand = random.Random(7)
    want = [int(i) for i in kept if rand.random() < 0.25]
    assert list(r.kept_indices()) == want
    r = BAMReader(d / "x.sam", downsample_to=100, downsample_seed=3)
    got = r.kept_indices()
    assert len(got) == 100 and len(set(got.tolist())) == 100
    keys = [(int(batch.tid[i]), int(batch.pos[i])) for i in got]
    assert keys == sorted(keys)


def test_downsampling_against_the_reference_golden():
    """tests/golden/downsample.npz: what the reference's own BAMReader._downsample_to_fraction / _downsample_to_fixed_number
    (reader.py:134-164) keep of seeded flag / tid / pos columns (tools/make_golden_downsample.py) — every case in one piece
    and, the fraction cases, in stretches that hand the generator on (the chunked and the device decode paths)."""
    import json

    from mapdamage_amd.reader import downsample_indices
    z = np.load(str(GOLDEN / "downsample.npz"))
    flag, tid, pos = z["flag"], z["tid"], z["pos"]
    for case in json.loads(bytes(z["cases"]).decode()):
        to, seed, want = case["downsample_to"], case["seed"], z[case["kept"]]
        got = downsample_indices(flag, tid, pos, to, random.Random(seed))
        np.testing.assert_array_equal(got, want, err_msg=str(case))
        if to < 1:
            rand, parts = random.Random(seed), []
            for lo in range(0, len(flag), 777):
                parts.append(lo + downsample_indices(flag[lo:lo + 777], tid[lo:lo + 777], pos[lo:lo + 777], to, rand))
            np.testing.assert_array_equal(np.concatenate(parts), want, err_msg="in stretches: %s" % case)


def test_statistics_mirror_writes_reference_format(tmp_path):
    g = Golden("config1_L70_A10_Q20")
    ts = oracle_tableset(g.ref, g.batch, g.libraries, g.length, g.around, g.minqual, lgd_max=4096)
    mis = MisincorporationRates.from_tables(ts)
    comp = DNAComposition.from_tables(ts)
    lgd = FragmentLengths.from_tables(ts)
    mis.write(tmp_path / "misincorporation.txt")
    comp.write(tmp_path / "dnacomp.txt")
    lgd.write(tmp_path / "lgdistribution.txt")
    for name in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt"):
        assert (tmp_path / name).read_text() == g.txt[name], name
    lib = ("Zed", "libB")
    li = ts.libraries.index(lib)
    assert mis.data[lib]["5p"]["+"]["C>T"][0] == int(ts.mis[li, 1, 0, 0, 5])
    assert comp.data[lib]["3p"]["-"]["G"][-1] == int(ts.comp[li, 0, 1, g.length - 1, 2])
    assert check_table_and_warn_if_dmg_freq_is_low(tmp_path) is True
    assert check_table_and_warn_if_dmg_freq_is_low(tmp_path / "missing") is False


def test_cli_parser_mirrors_reference_flags():
    from mapdamage_amd.main import build_parser
    p = build_parser()
    o = p.parse_args(["-i", "a.bam", "-r", "r.fa"])
    assert (o.length, o.around, o.minqual, o.readplot, o.refplot, o.ymax) == (70, 10, 0, 25, 10, 0.3)
    o = p.parse_args(["-i", "a.bam", "-r", "r.fa", "-l", "50", "-a", "5", "-Q", "20", "--merge-libraries",
                      "-n", "0.5", "--downsample-seed", "1", "--no-stats"])
    assert (o.length, o.around, o.minqual, o.merge_libraries, o.downsample) == (50, 5, 20, True, 0.5)
    with pytest.raises(SystemExit):
        p.parse_args(["-Q", "94"])
    with pytest.raises(SystemExit):
        p.parse_args(["-l", "0"])


def test_native_bam_decoder_matches_python_decoder(files):
    """mdx_bam_* (C++, multi-threaded inflate) against the pure-Python decoder, field by field."""
    d, ref, batch, rg_of = files
    py = sam.read_bam(d / "x.bam")
    for threads in (1, 4):
        nat = sam.read_bam_native(d / "x.bam", threads=threads)
        for k in ("flag", "tid", "pos", "tlen", "cigar_off", "cigar", "seq_off", "seq", "qual", "mtid", "mpos"):
            np.testing.assert_array_equal(getattr(nat.batch, k), getattr(py.batch, k), err_msg=k)
        assert nat.rg == py.rg == rg_of
        assert nat.qname == py.qname
        assert nat.header.references == ref.names and nat.header.lengths == ref.lengths
        assert nat.header.read_groups == py.header.read_groups


def test_native_bam_views_keep_the_decoder_alive(files):
    """The columns are views of the decoder's buffers: a slice must stay valid after the Alignments object and
    the batch it came from are gone."""
    import gc
    d, ref, batch, rg_of = files
    al = sam.read_bam_native(d / "x.bam")
    part = al.batch.slice(3, 40)
    seq, want = part.seq[5:200], bytes(batch.slice(3, 40).seq[5:200])
    del al, part
    gc.collect()
    junk = [np.ones(1 << 20, np.uint8) for _ in range(8)]    # churn the allocator
    del junk
    assert bytes(seq) == want


def test_native_bam_decoder_rejects_garbage(tmp_path):
    (tmp_path / "bad.bam").write_bytes(b"\x1f\x8bnot really a bam file at all")
    with pytest.raises(ValueError):
        sam.read_bam_native(tmp_path / "bad.bam")


def test_experimental_damage_frequency_files_follow_the_reference_table():
    """N3 (experimental, format unpinned): the numbers are the aggregation of the reference's own
    misincorporation.txt (sum over libraries and strands of C>T / C at the 5' end, G>A / G at 3')."""
    import csv
    import io

    from tests.util import Golden, oracle_tableset
    g = Golden("config1_L70_A10_Q0")
    ts = oracle_tableset(g.ref, g.batch, g.libraries, g.length, g.around, g.minqual)
    rows = list(csv.DictReader(io.StringIO(g.txt["misincorporation.txt"]), delimiter="\t"))
    for end, num, den, name in (("5p", "C>T", "C", "5pC>T"), ("3p", "G>A", "G", "3pG>A")):
        lines = ts.damage_frequency_text(end, 25).splitlines()
        assert lines[0] == "pos\t" + name and len(lines) == 26
        for p in range(1, 26):
            sel = [r for r in rows if r["End"] == end and int(r["Pos"]) == p]
            n, d = sum(int(r[num]) for r in sel), sum(int(r[den]) for r in sel)
            assert lines[p] == "%d\t%s" % (p, "%.15g" % (n / d))


def test_batch_take_and_view_slices():
    """Vectorised ``take`` equals the record-by-record gather; ``slice(copy=False)`` equals ``slice``."""
    from mapdamage_amd.batch import batch_from_records
    ref = synth.small_genome()
    b = synth.make_edge_reads(ref)
    idx = np.array([5, 0, 7, 7, 3, b.n - 1])
    got = b.take(idx).validate()
    want = batch_from_records([b.record(int(i)) for i in idx], with_qual=b.qual is not None)
    for k in ("flag", "lib", "tid", "pos", "tlen", "cigar_off", "cigar", "seq_off", "seq", "qual"):
        np.testing.assert_array_equal(getattr(got, k), getattr(want, k), err_msg=k)
    assert b.take(np.zeros(0, np.int64)).validate().n == 0
    a, v = b.slice(3, 17), b.slice(3, 17, copy=False).validate()
    for k in ("flag", "lib", "tid", "pos", "tlen", "cigar_off", "cigar", "seq_off", "seq", "qual"):
        np.testing.assert_array_equal(getattr(a, k), getattr(v, k), err_msg=k)


@pytest.mark.parametrize("chunk_bytes", [64, 1000, 4096, 50_000, 1 << 30])
def test_native_bam_stream_matches_whole_file_decode(files, chunk_bytes):
    """mdx_bam_open / mdx_bam_next: the chunks, concatenated, are the records of mdx_bam_read in file order,
    whatever the chunk size (records and BGZF blocks straddle chunk borders)."""
    from mapdamage_amd.batch import concat_batches
    d, ref, batch, rg_of = files
    whole = sam.read_bam_native(d / "x.bam")
    with sam.BamStream(d / "x.bam", threads=3, chunk_bytes=chunk_bytes) as stream:
        assert stream.header.references == ref.names and stream.header.lengths == ref.lengths
        assert stream.header.read_groups == whole.header.read_groups
        chunks = list(stream)
    if chunk_bytes < 50_000:
        assert len(chunks) > 3
    got = concat_batches([c.batch for c in chunks])
    for k in ("flag", "tid", "pos", "tlen", "cigar_off", "cigar", "seq_off", "seq", "qual"):
        np.testing.assert_array_equal(getattr(got, k), getattr(whole.batch, k), err_msg=k)
    assert [r for c in chunks for r in c.rg] == whole.rg == rg_of
    assert [q for c in chunks for q in c.qname] == whole.qname


def test_native_bam_stream_takes_a_file_up_in_the_middle(files, tmp_path):
    """mdx_bam_seek (what a device decode that gave up hands over, mdx_gbam_tell): the BGZF block at a compressed offset and
    the inflated bytes in front of the first record wanted — with records that straddle the blocks of the file."""
    import struct
    d, ref, batch, rg_of = files
    path = tmp_path / "cut.bam"
    sam.write_bam(path, batch, ref.names, ref.lengths, RGS, rg_of, htslib_blocks=False, block_bytes=700)
    whole = sam.read_bam_native(path)
    raw = path.read_bytes()
    # compressed offset and inflated size of every block; inflated offset of every record
    blocks, at, out = [], 0, 0
    while at < len(raw):
        size = struct.unpack_from("<H", raw, at + 16)[0] + 1
        isize = struct.unpack_from("<I", raw, at + size - 4)[0]
        blocks.append((at, out))
        out += isize
        at += size
    with sam.BamStream(path, threads=2, chunk_bytes=1 << 20, keep_raw=True) as stream:
        chunk = stream.next_chunk()
        assert chunk.batch.n == whole.batch.n
        import ctypes
        data, rec_off = ctypes.c_void_p(), ctypes.c_void_p()
        assert stream._lib.mdx_bam_raw(chunk.native, ctypes.byref(data), ctypes.byref(rec_off)) == 0
        offs = np.ctypeslib.as_array(ctypes.cast(rec_off, ctypes.POINTER(ctypes.c_uint64)), (whole.batch.n + 1,)).copy()
        header_bytes = out - int(offs[-1])
        for k in (1, whole.batch.n // 3, whole.batch.n - 2):
            start = header_bytes + int(offs[k])                 # inflated offset of record k
            b = max(i for i, (_, o) in enumerate(blocks) if o <= start)
            stream.seek(blocks[b][0], start - blocks[b][1])
            rest = stream.next_chunk()
            assert rest.batch.n == whole.batch.n - k
            np.testing.assert_array_equal(rest.batch.pos, whole.batch.pos[k:])
            np.testing.assert_array_equal(rest.batch.seq, whole.batch.seq[int(whole.batch.seq_off[k]):])
            assert rest.qname == whole.qname[k:]
            assert stream.next_chunk() is None
        with pytest.raises(ValueError):
            stream.seek(blocks[3][0] + 1, 0)                    # not a block boundary
            stream.next_chunk()


def test_native_bam_stream_rejects_garbage_and_truncation(files, tmp_path):
    d = files[0]
    (tmp_path / "bad.bam").write_bytes(b"\x1f\x8bnot really a bam file at all")
    with pytest.raises(ValueError):
        sam.BamStream(tmp_path / "bad.bam")
    data = (d / "x.bam").read_bytes()
    (tmp_path / "cut.bam").write_bytes(data[:len(data) // 2])
    with pytest.raises(ValueError):
        with sam.BamStream(tmp_path / "cut.bam", chunk_bytes=4096) as stream:
            list(stream)


@pytest.mark.parametrize("downsample", [None, 0.3])
def test_reader_in_chunks_yields_the_same_records(files, downsample):
    """BAMReader(chunk_bytes=...): the batches of a BAM file decoded in pieces, concatenated, are the batch
    of the one-piece decode — flag filter, library column and the --downsample stream of draws included."""
    from mapdamage_amd.batch import concat_batches
    d, ref, batch, rg_of = files
    (whole,) = list(BAMReader(d / "x.bam", downsample_to=downsample, downsample_seed=5).iter_batches())
    reader = BAMReader(d / "x.bam", downsample_to=downsample, downsample_seed=5, chunk_bytes=3000)
    assert reader.get_libraries() == [("Zed", "libB"), ("Alpha", "libA")]
    assert reader.get_references() == dict(zip(ref.names, ref.lengths))
    parts = list(reader.iter_batches())
    assert len(parts) > 5
    got = concat_batches(parts)
    for k in ("flag", "lib", "tid", "pos", "tlen", "cigar_off", "cigar", "seq_off", "seq", "qual"):
        np.testing.assert_array_equal(getattr(got, k), getattr(whole, k), err_msg=k)
    # a fixed-size sample and SAM text keep the one-piece decode
    assert len(list(BAMReader(d / "x.bam", downsample_to=50, chunk_bytes=3000).iter_batches())) == 1
    assert len(list(BAMReader(d / "x.sam", chunk_bytes=3000).iter_batches())) == 1


def test_reader_in_chunks_reports_readgroup_errors(files, tmp_path):
    d, ref, batch, rg_of = files
    bad = list(rg_of)
    bad[len(bad) // 2] = "nope"
    keep = np.nonzero((batch.flag & 0xF04) == 0)[0]
    bad[int(keep[len(keep) // 2])] = "nope"
    sam.write_bam(tmp_path / "bad.bam", batch, ref.names, ref.lengths, RGS, bad)
    with pytest.raises(BAMError, match="not listed in BAM header"):
        list(BAMReader(tmp_path / "bad.bam", chunk_bytes=3000).iter_batches())


def _decode_in_subprocess(path, chunk_bytes, scan_min):
    """Decoded columns (whole file or chunked) from a fresh interpreter with MDX_BAM_PARALLEL_SCAN_MIN set: the
    threshold is read once per process."""
    import json
    import os
    import subprocess
    import sys
    code = (
        "import sys, json, hashlib\n"
        "import numpy as np\n"
        "from mapdamage_amd import sam\n"
        "from mapdamage_amd.batch import concat_batches\n"
        "path, chunk = sys.argv[1], int(sys.argv[2])\n"
        "if chunk:\n"
        "    with sam.BamStream(path, threads=4, chunk_bytes=chunk) as st:\n"
        "        parts = list(st)\n"
        "    b = concat_batches([p.batch for p in parts]); names = [q for p in parts for q in p.qname]\n"
        "else:\n"
        "    al = sam.read_bam_native(path, threads=4); b = al.batch; names = al.qname\n"
        "h = hashlib.sha256()\n"
        "for k in ('flag', 'tid', 'pos', 'tlen', 'cigar_off', 'cigar', 'seq_off', 'seq', 'qual'):\n"
        "    h.update(np.ascontiguousarray(getattr(b, k)).tobytes())\n"
        "h.update('\\n'.join(names).encode())\n"
        "print(json.dumps({'n': b.n, 'sha': h.hexdigest()}))\n")
    env = dict(os.environ, MDX_BAM_PARALLEL_SCAN_MIN=str(scan_min))
    out = subprocess.run([sys.executable, "-c", code, str(path), str(chunk_bytes)], env=env, check=True,
                         capture_output=True, text=True, cwd=str(__import__("pathlib").Path(__file__).parent.parent))
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("htslib_blocks", [True, False])
def test_native_bam_parallel_record_scan_equals_serial_scan(files, tmp_path, htslib_blocks):
    """The speculative parallel record scan (segments started at BGZF block starts, accepted only when every
    chain lands on the next start) against the serial scan: equal columns when the speculation holds (htslib
    block layout) and when it cannot (records straddling blocks -> serial fallback), whole file and chunked."""
    d, ref, batch, rg_of = files
    path = tmp_path / "layout.bam"
    sam.write_bam(path, batch, ref.names, ref.lengths, RGS, rg_of, htslib_blocks=htslib_blocks)
    want = _decode_in_subprocess(path, 0, 1 << 40)            # serial scan
    assert want["n"] == batch.n
    for chunk in (0, 70_000, 5_000):
        assert _decode_in_subprocess(path, chunk, 0) == want, chunk


def test_native_decoder_rejects_hostile_bgzf_sizes(tmp_path):
    """ISIZE is read from the file — 70 000 tiny blocks claiming 4 GiB each must be refused (not
    allocated), a subfield may not run past the extra field, and no C++ exception may leave the C boundary."""
    import struct
    import zlib

    from mapdamage_amd import sam

    def block(payload, isize=None, extra=None):
        comp = zlib.compressobj(6, zlib.DEFLATED, -15)
        data = comp.compress(payload) + comp.flush()
        if extra is None:
            extra = b"BC" + struct.pack("<HH", 2, 12 + 6 + len(data) + 8 - 1)
        hdr = b"\x1f\x8b\x08\x04" + b"\0" * 6 + struct.pack("<H", len(extra)) + extra
        return hdr + data + struct.pack("<II", zlib.crc32(payload), len(payload) if isize is None else isize)

    huge = tmp_path / "huge_isize.bam"
    huge.write_bytes(b"".join(block(b"x", isize=0xFFFFFFFF) for _ in range(70000)))
    with pytest.raises((ValueError, sam.BAMError), match="BGZF"):
        sam.read_bam_native(huge)
    with pytest.raises((ValueError, sam.BAMError), match="BGZF"):
        sam.BamStream(huge)
    # a subfield whose declared length runs past the extra field (and past the end of the file)
    odd = tmp_path / "subfield.bam"
    odd.write_bytes(b"\x1f\x8b\x08\x04" + b"\0" * 6 + struct.pack("<H", 6) + b"XY" + struct.pack("<H", 60000) + b"\0\0")
    with pytest.raises((ValueError, sam.BAMError)):
        sam.read_bam_native(odd)


def test_native_patch_of_rescaled_records_and_bgzf_writer(tmp_path):
    """`--rescale-only` writes every record back: the native decoder keeps the encoded records of a chunk, the
    patch replaces QUAL and appends MR:f on the flagged ones and leaves every other byte alone; the threaded BGZF
    writer produces a file the decoders read back."""
    import struct

    from mapdamage_amd import sam
    ref, batch = synth.config1_batch()
    path = tmp_path / "in.bam"
    sam.write_bam(path, batch, ref.names, ref.lengths, RGS, ["rgA" if i % 2 else "rgB" for i in range(batch.n)])
    whole = sam.read_bam(path, keep_raw=True)
    rng = np.random.default_rng(1)
    with sam.BamStream(path, chunk_bytes=40_000, keep_raw=True) as stream, sam.BgzfWriter(tmp_path / "out.bam", threads=3) as out:
        out.write(sam.bam_header_bytes(stream.header))
        first, picked, new_q, mrs = 0, [], [], []
        for chunk in stream:
            b = chunk.batch
            flags = (rng.random(b.n) < 0.4).astype(np.uint8)
            qual_out = rng.integers(0, 42, b.seq.shape[0]).astype(np.uint8)
            mr = rng.random(b.n).astype(np.float32)
            # nothing flagged: the records come back byte for byte
            same = stream.patch_rescaled(chunk, qual_out, mr, np.zeros(b.n, np.uint8))
            want = b"".join(struct.pack("<i", len(r)) + r for r in whole.raw[first:first + b.n])
            assert same.tobytes() == want
            out.write(stream.patch_rescaled(chunk, qual_out, mr, flags))
            for i in np.nonzero(flags)[0]:
                picked.append(first + int(i))
                new_q.append(qual_out[int(b.seq_off[i]):int(b.seq_off[i + 1])].copy())
                mrs.append(mr[i])
            first += b.n
    assert first == batch.n and len(picked) > 100
    back = sam.read_bam(tmp_path / "out.bam", keep_raw=True)
    assert back.batch.n == batch.n and back.header.text == whole.header.text
    np.testing.assert_array_equal(back.batch.seq, batch.seq)
    chosen = set(picked)
    for k, i in enumerate(picked):
        s0, s1 = int(batch.seq_off[i]), int(batch.seq_off[i + 1])
        np.testing.assert_array_equal(back.batch.qual[s0:s1], new_q[k])
        assert back.raw[i][-7:-4] == b"MRf" and struct.unpack("<f", back.raw[i][-4:])[0] == mrs[k]
    for i in range(batch.n):
        if i not in chosen:
            assert back.raw[i] == whole.raw[i]


def test_bgzf_block_crc_is_checked(tmp_path):
    """A block whose bytes inflate but do not match its CRC32 is refused (htslib, behind the reference's pysam, does
    the same), by the one-piece and by the chunked decoder."""
    import struct

    import pytest

    from mapdamage_amd import sam, synth
    ref = synth.make_genome(seed=2, sizes=(("c1", 50_000),), n_run=50, lower_run=300)
    b = synth.make_reads(ref, 3000, 3, len_range=(30, 100))
    path = tmp_path / "x.bam"
    sam.write_bam(str(path), b, ref.names, ref.lengths, [{"ID": "r", "SM": "s", "LB": "l"}], rg_of_record=["r"] * b.n)
    raw = bytearray(path.read_bytes())
    # walk to the third block and damage its CRC32 field (the DEFLATE stream and ISIZE stay valid)
    off = 0
    for _ in range(2):
        off += struct.unpack_from("<H", raw, off + 16)[0] + 1
    bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
    raw[off + bsize - 8] ^= 0x01
    bad = tmp_path / "bad.bam"
    bad.write_bytes(bytes(raw))
    assert sam.read_bam_native(str(path)).batch.n == b.n
    with pytest.raises(ValueError, match="CRC32"):
        sam.read_bam_native(str(bad))
    with pytest.raises(ValueError, match="CRC32"):
        with sam.BamStream(str(bad), chunk_bytes=1 << 16) as stream:
            for _ in stream:
                pass


def test_decoders_clear_flag_bit_15(files, tmp_path):
    """Bit 15 of a FLAG field is the kernel's MDX_FLAG_QUAL_ABOVE_MIN hint (include/mdx.h), never the file's: the SAM
    text parser and both BAM decoders clear it; mark_unmaskable decides it from the qualities alone."""
    from mapdamage_amd.batch import mark_unmaskable
    d, ref, batch, rg_of = files
    dirty = batch.slice(0, batch.n)
    dirty.flag = (dirty.flag | np.uint16(0x8000)).astype(np.uint16)
    sam.write_sam(tmp_path / "d.sam", dirty, ref.names, ref.lengths, RGS, rg_of)
    sam.write_bam(tmp_path / "d.bam", dirty, ref.names, ref.lengths, RGS, rg_of)
    assert (sam.read_sam(tmp_path / "d.sam").batch.flag == batch.flag).all()
    assert (sam.read_bam(tmp_path / "d.bam").batch.flag == batch.flag).all()
    assert (sam.read_bam_native(tmp_path / "d.bam").batch.flag == batch.flag).all()
    marked, nothing = mark_unmaskable(dirty, 94)          # every quality is below 94: nothing may keep the bit
    has_q = np.asarray([batch.qual[batch.seq_off[i]] != 0xFF if batch.seq_off[i + 1] > batch.seq_off[i] else False
                        for i in range(batch.n)])
    assert not (marked.flag[has_q] & 0x8000).any() and not nothing


def test_cli_gpus_prints_the_launch_it_would_re_execute_itself_under(tmp_path, capsys):
    """`--gpus 8 --print-launch`: one rank per GPU under torch.distributed.run on 127.0.0.1, the command's own
    arguments behind it (no GPU needed: nothing is launched)."""
    import json

    from mapdamage_amd.main import main
    argv = ["-i", str(tmp_path / "x.bam"), "-r", str(tmp_path / "ref.fa"), "-d", str(tmp_path / "out"), "--gpus", "8",
            "--print-launch"]
    assert main(argv) == 0
    cmd = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index("mapdamage_amd") + 1:]
    assert tail == [a for a in argv if a != "--print-launch"] and cmd[cmd.index("mapdamage_amd") - 1] == "-m"
