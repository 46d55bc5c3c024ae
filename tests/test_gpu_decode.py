"""GPU-side BAM decode (include/mdx.h mdx_gbam_*): the columns the device produces equal the host decoder's, and
the tables counted from them equal the oracle's."""
import ctypes

import numpy as np
import pytest

from mapdamage_amd import sam, synth

pytestmark = pytest.mark.gpu

RGS = [{"ID": "rgA", "SM": "s", "LB": "lib1"}, {"ID": "rg_b2", "SM": "s", "LB": "lib2"}, {"ID": "x", "SM": "s", "LB": "lib1"}]


# how the BGZF blocks of a test file are laid out: as htslib does (every block starts at a record, the header in blocks of
# its own), or header and records as one stream cut anywhere — 0xFF00 bytes, htsjdk's 65 498, and blocks so small that a
# record spans several of them
LAYOUTS = {"htslib": dict(), "cut": dict(htslib_blocks=False), "htsjdk": dict(htslib_blocks=False, block_bytes=65498),
           "tiny": dict(htslib_blocks=False, block_bytes=90)}


def _write(tmp_path, n=30_000, seed=4, layout="htslib", **kw):
    ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500, lower_run=3000)
    b = synth.make_reads(ref, n, seed, len_range=(25, 160), paired=True, frac_softclip=0.2, frac_ins=0.08, frac_del=0.08,
                         frac_skip=0.01, with_qual=True, frac_filtered=0.05, **kw)
    rng = np.random.default_rng(seed)
    rg = [RGS[i]["ID"] for i in rng.integers(0, 3, size=b.n)]
    path = tmp_path / "g.bam"
    sam.write_bam(str(path), b, ref.names, ref.lengths, RGS, rg_of_record=rg, **LAYOUTS[layout])
    return ref, b, rg, path


def _d2h(ptr, n, dtype):
    """n elements at device address `ptr` (hipMemcpy through the HIP runtime the library itself is linked to)."""
    out = np.empty(n, dtype)
    if n:
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        assert hip.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2) == 0        # hipMemcpyDeviceToHost
    return out


@pytest.mark.parametrize("chunk,layout", [(1 << 16, "htslib"), (1 << 20, "htslib"), (1 << 28, "htslib"), (1 << 16, "cut"),
                                          (1 << 28, "cut"), (1 << 18, "htsjdk"), (1 << 16, "tiny")])
def test_device_columns_equal_host_decoder(tmp_path, chunk, layout):
    """... whatever the BGZF layout: records that straddle blocks and slabs, a header that shares its block with records."""
    from mapdamage_amd.engine import DamageEngine
    ref, b, rg, path = _write(tmp_path, n=3_000 if layout == "tiny" else 30_000, layout=layout)
    host = sam.read_bam_native(str(path))
    hb = host.batch
    lib_of = {"rgA": 0, "rg_b2": 1, "x": 0}
    want_lib = np.asarray([lib_of[host.rg_names[i]] for i in host.rg_index], np.uint16)
    with DamageEngine([("s", "lib1"), ("s", "lib2")]) as eng:
        eng.set_reference(ref)
        with sam.GpuBamStream(eng, str(path), readgroups=list(lib_of.items()), chunk_bytes=chunk, want_qual=True, want_mate=True) as g:
            assert g.header.references == host.header.references
            got = {k: [] for k in ("flag", "lib", "tid", "pos", "tlen", "cigar", "seq", "qual", "mtid", "mpos", "clen", "slen")}
            n = 0
            while True:
                v = g.next_view()
                if v is None:
                    break
                eng.sync()
                k = int(v.n_reads)
                n += k
                got["flag"].append(_d2h(v.flag, k, np.uint16)); got["lib"].append(_d2h(v.lib, k, np.uint16))
                for name in ("tid", "pos", "tlen", "mtid", "mpos"):
                    got[name].append(_d2h(getattr(v, name), k, np.int32))
                co = _d2h(v.cigar_off, k + 1, np.uint32); so = _d2h(v.seq_off, k + 1, np.uint32)
                assert co[0] == 0 and so[0] == 0 and co[-1] == v.n_cigar and so[-1] == v.n_bases
                got["clen"].append(np.diff(co)); got["slen"].append(np.diff(so))
                got["cigar"].append(_d2h(v.cigar, int(v.n_cigar), np.uint32))
                got["seq"].append(_d2h(v.seq, int(v.n_bases), np.uint8)); got["qual"].append(_d2h(v.qual, int(v.n_bases), np.uint8))
    cat = {k: np.concatenate(v) for k, v in got.items()}
    assert n == hb.n
    # (the device decoder's flag column carries the hint MDX_FLAG_HAS_QUAL, 0x4000, on the records that have qualities — at least
    # one base and a first quality byte that is not 0xFF, main.py:185 / rescale.py:306 — on top of the file's bits)
    lens = np.diff(hb.seq_off.astype(np.int64))
    first = hb.qual[np.minimum(hb.seq_off[:-1].astype(np.int64), max(0, hb.qual.shape[0] - 1))] if hb.qual.shape[0] else np.zeros(hb.n, np.uint8)
    has_qual = (lens > 0) & (first != 0xFF)
    np.testing.assert_array_equal((cat["flag"] & 0x4000) != 0, has_qual)
    cat["flag"] = cat["flag"] & np.uint16(0x3FFF)
    for name in ("flag", "tid", "pos", "tlen", "cigar", "seq", "qual"):
        np.testing.assert_array_equal(cat[name], getattr(hb, name), err_msg=name)
    np.testing.assert_array_equal(cat["mtid"], hb.mtid); np.testing.assert_array_equal(cat["mpos"], hb.mpos)
    np.testing.assert_array_equal(cat["clen"], np.diff(hb.cigar_off)); np.testing.assert_array_equal(cat["slen"], np.diff(hb.seq_off))
    np.testing.assert_array_equal(cat["lib"], want_lib)


@pytest.mark.parametrize("chunk", [1 << 16, 1 << 28])
def test_device_seq_column_in_its_4bit_form(tmp_path, chunk):
    """MDX_SEQ_4BIT from the device decode path: BAM's nibbles kept as nibbles — recoded, low nibble first, records of odd
    length followed by the next record's first base in the same byte — equal the host packer over the host decoder's SEQ."""
    from mapdamage_amd.engine import SEQ_4BIT, DamageEngine, pack_seq
    ref, b, rg, path = _write(tmp_path, frac_n_base=0.03)
    hb = sam.read_bam_native(str(path)).batch
    assert (hb.seq == ord("N")).any()
    assert (np.diff(hb.seq_off.astype(np.int64)) & 1).any()          # records of odd length among them
    with DamageEngine([("s", "lib1"), ("s", "lib2")]) as eng:
        eng.set_reference(ref)
        with sam.GpuBamStream(eng, str(path), readgroups=[("rgA", 0), ("rg_b2", 1), ("x", 0)], chunk_bytes=chunk) as g:
            assert g.packed
            at = 0
            while True:
                v = g.next_view()
                if v is None:
                    break
                eng.sync()
                assert v.seq_format == SEQ_4BIT
                k, nb = int(v.n_reads), int(v.n_bases)
                so = _d2h(v.seq_off, k + 1, np.uint32)
                s0 = int(hb.seq_off[at])
                np.testing.assert_array_equal(so, hb.seq_off[at:at + k + 1] - np.uint32(s0))
                got = _d2h(v.seq, (nb + 1) // 2, np.uint8)
                np.testing.assert_array_equal(got, pack_seq(hb.seq[s0:s0 + nb]))
                at += k
            assert at == hb.n


def test_tables_from_device_decode_equal_oracle(tmp_path):
    from mapdamage_amd.engine import DamageEngine
    from oracle import oracle
    ref, b, rg, path = _write(tmp_path, n=60_000, seed=9)
    lib_of = {"rgA": 0, "rg_b2": 1, "x": 0}
    host = sam.read_bam_native(str(path))
    hb = host.batch
    hb.lib = np.asarray([lib_of[host.rg_names[i]] for i in host.rg_index], np.uint16)
    want = oracle.tabulate(ref, hb, 2, 70, 10)
    with DamageEngine([("s", "lib1"), ("s", "lib2")]) as eng:
        eng.set_reference(ref)
        with sam.GpuBamStream(eng, str(path), readgroups=list(lib_of.items()), chunk_bytes=1 << 19) as g:
            while True:
                v = g.next_view()
                if v is None:
                    break
                eng.tabulate_view(v)
        got = eng.finish()
    np.testing.assert_array_equal(got.mis, want["mis"])
    np.testing.assert_array_equal(got.comp, want["comp"])
    assert got.n_kept == want["n_kept"]


@pytest.mark.parametrize("golden,extra", [("config1_L70_A10_Q0", []), ("config1_merged_L70_A10_Q0", ["--merge-libraries"]),
                                          ("config1_L70_A10_Q20", ["-Q", "20"]), ("indelshapes_L70_A10_Q20", ["-Q", "20"]),
                                          ("indelshapes_L70_A10_Q0", []), ("edge_L70_A10_Q0", []), ("config4s_L70_A10", [])])
def test_cli_gpu_decode_writes_the_reference_tables(tmp_path, golden, extra):
    """`python -m mapdamage_amd --gpu-decode`: the three tables of the reference, byte for byte, from a file that was
    inflated and unpacked on the GPU (several slabs)."""
    import pathlib

    from mapdamage_amd import fasta
    from mapdamage_amd.main import main
    from tests.util import Golden
    g = Golden(golden)
    rgs = [{"ID": "rg%d" % i, "SM": s, "LB": l} for i, (s, l) in enumerate(g.meta["libraries"])]
    raw_lib = np.load(str(pathlib.Path(__file__).parent / "golden" / (golden + ".npz")))["lib"]
    if "--merge-libraries" in extra:
        rgs = [{"ID": "rg0", "SM": "a", "LB": "b"}, {"ID": "rg1", "SM": "c", "LB": "d"}]
        rg_of = ["rg%d" % (i % 2) for i in range(g.batch.n)]
    else:
        rg_of = ["rg%d" % int(l) for l in raw_lib]
    path = tmp_path / "in.bam"
    sam.write_bam(path, g.batch, g.ref.names, g.ref.lengths, rgs, rg_of)
    fasta.write_fasta(tmp_path / "ref.fa", g.ref)
    for chunk_mb in ("1024", "0.3"):
        out = tmp_path / ("res" + chunk_mb)
        assert main(["-i", str(path), "-r", str(tmp_path / "ref.fa"), "-d", str(out), "--no-stats", "--gpu-decode",
                     "--chunk-mb", chunk_mb] + extra) == 0
        for name in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt"):
            assert (out / name).read_text() == g.txt[name], (chunk_mb, name)
        assert "decoding on the host" not in (out / "Runtime_log.txt").read_text()


@pytest.mark.parametrize("layout", ["cut", "htsjdk", "tiny"])
def test_any_bgzf_layout_on_the_device_path(tmp_path, layout):
    """Records straddling BGZF blocks, a header sharing its block with records (htsjdk / Picard, sambamba, biobambam): the
    device path takes the file — no fallback — and the command line writes what the host decoder writes."""
    from mapdamage_amd import fasta
    from mapdamage_amd.main import main
    ref, b, rg, path = _write(tmp_path, n=2_000 if layout == "tiny" else 20_000, layout=layout)
    fasta.write_fasta(tmp_path / "ref.fa", ref)
    outs = []
    for name, flags in (("host", ["--host-decode"]), ("dev", ["--gpu-decode", "--chunk-mb", "1"])):
        out = tmp_path / name
        assert main(["-i", str(path), "-r", str(tmp_path / "ref.fa"), "-d", str(out), "--no-stats", "--log-level", "DEBUG"] + flags) == 0
        outs.append([(out / f).read_text() for f in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt")])
    assert outs[0] == outs[1]
    log = (tmp_path / "dev" / "Runtime_log.txt").read_text()
    assert "decoding on the host" not in log and "gave up" not in log


def test_the_hosts_share_of_the_inflate(tmp_path):
    """The last part of a slab's BGZF blocks inflated by host threads into pinned memory and copied to their place in HBM
    under the device's inflate of the rest (forced on a small file: MDX_GBAM_HOST_SHARE / _MIN_BLOCKS, read when the library
    first decodes — hence a process of its own): same columns as the host decoder for both block layouts, and a damaged
    block in the host's part is the error it is in the device's."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys
        import numpy as np
        sys.path.insert(0, %r)
        from tests import test_gpu_decode as t
        from mapdamage_amd import sam
        from mapdamage_amd.engine import DamageEngine
        import pathlib, tempfile
        for layout in ("htslib", "cut"):
            d = pathlib.Path(tempfile.mkdtemp())
            ref, b, rg, path = t._write(d, n=30000, layout=layout)
            host = sam.read_bam_native(str(path))
            with DamageEngine([("s", "lib1"), ("s", "lib2")]) as eng:
                eng.set_reference(ref)
                with sam.GpuBamStream(eng, str(path), readgroups=[("rgA", 0), ("rg_b2", 1), ("x", 0)], chunk_bytes=1 << 20,
                                      want_qual=True) as g:
                    pos, seq, qual, n = [], [], [], 0
                    while (v := g.next_view()) is not None:
                        eng.sync()
                        k = int(v.n_reads); n += k
                        pos.append(t._d2h(v.pos, k, np.int32)); seq.append(t._d2h(v.seq, int(v.n_bases), np.uint8))
                        qual.append(t._d2h(v.qual, int(v.n_bases), np.uint8))
                assert n == host.batch.n
                np.testing.assert_array_equal(np.concatenate(pos), host.batch.pos)
                np.testing.assert_array_equal(np.concatenate(seq), host.batch.seq)
                np.testing.assert_array_equal(np.concatenate(qual), host.batch.qual)
                # a flipped byte near the end of the file: in the host's share of the last slab
                raw = bytearray(path.read_bytes())
                raw[len(raw) - 3000] ^= 0x5A
                broken = d / "broken.bam"
                broken.write_bytes(bytes(raw))
                try:
                    with sam.GpuBamStream(eng, str(broken), readgroups=[("rgA", 0), ("rg_b2", 1), ("x", 0)], chunk_bytes=1 << 20) as g:
                        while g.next_view() is not None:
                            pass
                    raise SystemExit("a damaged block went unnoticed")
                except ValueError as e:
                    assert "corrupt BGZF block" in str(e), e
        print("host share ok")
    """ % root)
    env = dict(os.environ, MDX_GBAM_HOST_SHARE="0.5", MDX_GBAM_HOST_MIN_BLOCKS="8", MDX_GBAM_HOST_THREADS="6")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "host share ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_a_damaged_block_is_an_error_on_the_device_path(tmp_path):
    from mapdamage_amd.engine import DamageEngine
    ref, b, rg, path = _write(tmp_path, n=5000)
    with DamageEngine([("s", "lib1"), ("s", "lib2")]) as eng:
        eng.set_reference(ref)
        # a flipped byte in the middle of a block's payload
        raw = bytearray(path.read_bytes())
        raw[len(raw) // 2] ^= 0x5A
        broken = tmp_path / "broken.bam"
        broken.write_bytes(bytes(raw))
        with pytest.raises(ValueError):
            with sam.GpuBamStream(eng, str(broken), readgroups=[("rgA", 0), ("rg_b2", 1), ("x", 0)]) as g:
                while (v := g.next_view()) is not None:
                    eng.tabulate_view(v)
                eng.sync()
    # a file that ends inside a record
    cut = tmp_path / "cut.bam"
    sam.write_bam(str(cut), b, ref.names, ref.lengths, RGS, rg_of_record=rg, htslib_blocks=False)
    whole = cut.read_bytes()
    stream = sam.read_bam_native(str(cut))          # (sanity: the file is fine as written)
    assert stream.batch.n == b.n
    # drop the last data block (keep the EOF marker): the record that straddled into it is incomplete
    import struct
    offs, at = [], 0
    while at < len(whole):
        offs.append(at)
        at += struct.unpack_from("<H", whole, at + 16)[0] + 1
    short = tmp_path / "short.bam"
    short.write_bytes(whole[:offs[-2]] + whole[offs[-1]:])
    with DamageEngine([("s", "lib1"), ("s", "lib2")]) as eng:
        eng.set_reference(ref)
        with pytest.raises(ValueError, match="incomplete|truncated"):
            with sam.GpuBamStream(eng, str(short), readgroups=[("rgA", 0), ("rg_b2", 1), ("x", 0)], chunk_bytes=1 << 16) as g:
                while g.next_view() is not None:
                    pass


def test_the_fallback_goes_on_where_the_device_path_stopped(tmp_path, monkeypatch):
    """A slab the device path gives up on (forced: MDX_GBAM_FAIL_AT): the slabs in front of it stay counted, the host
    decoder takes the file up at that slab's first record — a record that straddles the slab border included — and the
    tables are those of a run on either path alone; the log says so."""
    from mapdamage_amd import fasta
    from mapdamage_amd.main import main
    for layout in ("htslib", "cut"):
        d = tmp_path / layout
        d.mkdir()
        ref, b, rg, path = _write(d, n=40_000, layout=layout)
        fasta.write_fasta(d / "ref.fa", ref)
        outs = {}
        for name, flags, fail in (("host", ["--host-decode"], None), ("dev", ["--gpu-decode", "--chunk-mb", "1"], None),
                                  ("resumed", ["--gpu-decode", "--chunk-mb", "1"], "3")):
            if fail is None:
                monkeypatch.delenv("MDX_GBAM_FAIL_AT", raising=False)
            else:
                monkeypatch.setenv("MDX_GBAM_FAIL_AT", fail)
            out = d / name
            assert main(["-i", str(path), "-r", str(d / "ref.fa"), "-d", str(out), "--no-stats", "--log-level", "DEBUG"] + flags) == 0
            outs[name] = [(out / f).read_text() for f in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt")]
        monkeypatch.delenv("MDX_GBAM_FAIL_AT", raising=False)
        assert outs["host"] == outs["dev"] == outs["resumed"], layout
        log = (d / "resumed" / "Runtime_log.txt").read_text()
        assert "WARNING GPU decode path gave up" in log and "MDX_ERR_UNSUPPORTED" in log
        assert "from compressed offset" in log and "the whole file again" not in log
        assert "WARNING Decode path: host decoder; fallbacks from the device path: 1" in log


def test_empty_file_and_records_without_read_group(tmp_path):
    """A BAM that holds a header and nothing else; records without RG tag: counted under --merge-libraries, the
    reference's BAMError (through the host path, which names the read) otherwise."""
    from mapdamage_amd import fasta
    from mapdamage_amd.batch import ReadBatch
    from mapdamage_amd.engine import DamageEngine
    from mapdamage_amd.main import main
    from mapdamage_amd.sam import BAMError
    ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500, lower_run=3000)
    empty = ReadBatch(np.zeros(0, np.uint16), np.zeros(0, np.uint16), np.zeros(0, np.int32), np.zeros(0, np.int32),
                      np.zeros(0, np.int32), np.zeros(1, np.uint32), np.zeros(0, np.uint32), np.zeros(1, np.uint32),
                      np.zeros(0, np.uint8), np.zeros(0, np.uint8))
    p0 = tmp_path / "empty.bam"
    sam.write_bam(str(p0), empty, ref.names, ref.lengths, RGS, rg_of_record=[])
    with DamageEngine([("s", "lib1"), ("s", "lib2")]) as eng:
        eng.set_reference(ref)
        with sam.GpuBamStream(eng, str(p0), readgroups=[("rgA", 0), ("rg_b2", 1), ("x", 0)]) as g:
            n = 0
            while (v := g.next_view()) is not None:
                n += int(v.n_reads)
                eng.tabulate_view(v)
            assert n == 0
        assert eng.finish().n_kept == 0
    b = synth.make_reads(ref, 3000, 5, len_range=(30, 120), frac_softclip=0.1)
    p1 = tmp_path / "norg.bam"
    sam.write_bam(str(p1), b, ref.names, ref.lengths, RGS, rg_of_record=None)
    fasta.write_fasta(tmp_path / "ref.fa", ref)
    outs = []
    for name, flags in (("host", ["--host-decode"]), ("dev", ["--gpu-decode"])):
        out = tmp_path / name
        assert main(["-i", str(p1), "-r", str(tmp_path / "ref.fa"), "-d", str(out), "--no-stats", "--merge-libraries"] + flags) == 0
        outs.append([(out / f).read_text() for f in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt")])
    assert outs[0] == outs[1]
    assert "decoding on the host" not in (tmp_path / "dev" / "Runtime_log.txt").read_text()
    with pytest.raises(BAMError, match="has no read-group"):
        main(["-i", str(p1), "-r", str(tmp_path / "ref.fa"), "-d", str(tmp_path / "err"), "--no-stats", "--gpu-decode"])


def test_damaged_files_end_in_an_error_not_a_hang(tmp_path):
    """Sixty copies of a small BAM with a few bytes changed somewhere behind the header: the device path either
    reports the damage (a corrupt block, records that do not add up) or — a change the DEFLATE stream and ISIZE
    survive, the block CRC32 is not checked — returns columns; it never hangs or faults."""
    from mapdamage_amd.engine import BadReadError, DamageEngine
    ref, b, rg, path = _write(tmp_path, n=4000)
    raw = path.read_bytes()
    rng = np.random.default_rng(77)
    outcomes = {"error": 0, "unsupported": 0, "decoded": 0}
    with DamageEngine([("s", "lib1"), ("s", "lib2")]) as eng:
        eng.set_reference(ref)
        for k in range(60):
            data = bytearray(raw)
            for _ in range(int(rng.integers(1, 4))):
                i = int(rng.integers(2000, len(data) - 28))
                data[i] ^= int(rng.integers(1, 256))
            bad = tmp_path / ("bad%d.bam" % k)
            bad.write_bytes(bytes(data))
            try:
                with sam.GpuBamStream(eng, str(bad), readgroups=[("rgA", 0), ("rg_b2", 1), ("x", 0)], chunk_bytes=1 << 18) as g:
                    while (v := g.next_view()) is not None:
                        eng.tabulate_view(v)
                    eng.sync()
                outcomes["decoded"] += 1
            except sam.GpuDecodeUnsupported:
                outcomes["unsupported"] += 1
            except (ValueError, BadReadError):
                outcomes["error"] += 1
            eng.reset()
            bad.unlink()
    assert outcomes["error"] + outcomes["unsupported"] > 30, outcomes


def test_rescaling_from_device_decoded_columns(tmp_path):
    """Qualities and mate columns unpacked on the GPU feed `mdx_tabulate_rescale_device` directly: the rescaled
    qualities, MR sums and routing equal those of the host-decoded batch, and the tables equal the oracle's."""
    import types

    import torch

    from mapdamage_amd.engine import DamageEngine
    from mapdamage_amd.rescale import RescaleModel
    from oracle import oracle
    ref, b, rg, path = _write(tmp_path, n=20_000, seed=6)
    rng = np.random.default_rng(5)
    corr_prob = {}
    for p in list(range(1, 13)) + list(range(-12, 0)):
        corr_prob[("C", "T", p)] = float(rng.random() * 0.5)
        corr_prob[("G", "A", p)] = float(rng.random() * 0.5)
    model = RescaleModel(corr_prob, 12, 12)
    host = sam.read_bam_native(str(path))
    hb = host.batch
    lib_of = {"rgA": 0, "rg_b2": 1, "x": 0}
    hb.lib = np.asarray([lib_of[host.rg_names[i]] for i in host.rg_index], np.uint16)
    want_tables = oracle.tabulate(ref, hb, 2, 70, 10)
    with DamageEngine([("s", "lib1"), ("s", "lib2")]) as eng:
        eng.set_reference(ref)
        eng.set_rescale_model(model)
        want_q, want_mr, want_st = eng.rescale(hb)
        eng.reset()
        got_q, got_mr, got_st = [], [], []
        with sam.GpuBamStream(eng, str(path), readgroups=list(lib_of.items()), chunk_bytes=1 << 19, want_qual=True, want_mate=True) as g:
            while (v := g.next_view()) is not None:
                n, nb = int(v.n_reads), int(v.n_bases)
                q_out = torch.empty(max(nb, 1), dtype=torch.uint8, device="cuda")
                mr = torch.empty(max(n, 1), dtype=torch.float64, device="cuda")
                st = torch.empty(max(n, 1), dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()
                eng.rescale_device(types.SimpleNamespace(dev=v), v.mtid, v.mpos, q_out.data_ptr(), mr.data_ptr(), st.data_ptr(),
                                   with_tables=True)
                eng.sync()
                got_q.append(q_out.cpu().numpy()[:nb]); got_mr.append(mr.cpu().numpy()[:n]); got_st.append(st.cpu().numpy()[:n])
        tables = eng.finish()
    got_q, got_mr, got_st = np.concatenate(got_q), np.concatenate(got_mr), np.concatenate(got_st)
    np.testing.assert_array_equal(got_q, want_q)
    np.testing.assert_array_equal(got_st, want_st)
    assert np.array_equal(np.isnan(got_mr), np.isnan(want_mr))
    np.testing.assert_array_equal(got_mr[~np.isnan(got_mr)], want_mr[~np.isnan(want_mr)])
    np.testing.assert_array_equal(tables.mis, want_tables["mis"])
    np.testing.assert_array_equal(tables.comp, want_tables["comp"])


def test_block_crc_is_checked_on_the_device(tmp_path):
    """A block whose DEFLATE stream and ISIZE are intact but whose CRC32 field is not: an error, as on the host."""
    import struct

    from mapdamage_amd.engine import DamageEngine
    ref, b, rg, path = _write(tmp_path, n=12000)
    raw = bytearray(path.read_bytes())
    off = 0
    for _ in range(30):          # (a block behind the first megabyte, which the header parse on the host inflates)
        off += struct.unpack_from("<H", raw, off + 16)[0] + 1
    bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
    raw[off + bsize - 7] ^= 0x40
    bad = tmp_path / "badcrc.bam"
    bad.write_bytes(bytes(raw))
    with DamageEngine([("s", "lib1"), ("s", "lib2")]) as eng:
        eng.set_reference(ref)
        with pytest.raises(ValueError, match="CRC32"):
            with sam.GpuBamStream(eng, str(bad), readgroups=[("rgA", 0), ("rg_b2", 1), ("x", 0)]) as g:
                while g.next_view() is not None:
                    pass
        # (and the intact file passes the same check)
        with sam.GpuBamStream(eng, str(path), readgroups=[("rgA", 0), ("rg_b2", 1), ("x", 0)], chunk_bytes=1 << 17) as g:
            n = 0
            while (v := g.next_view()) is not None:
                n += int(v.n_reads)
        assert n == b.n
    # the command line: the device path reports the damage, the host decoder has the last word
    from mapdamage_amd import fasta
    from mapdamage_amd.main import main
    fasta.write_fasta(tmp_path / "ref.fa", ref)
    with pytest.raises(ValueError, match="CRC32"):
        main(["-i", str(bad), "-r", str(tmp_path / "ref.fa"), "-d", str(tmp_path / "out"), "--no-stats"])
    assert "GPU decode path" in (tmp_path / "out" / "Runtime_log.txt").read_text()


def test_min_basequal_on_the_device_path(tmp_path):
    """-Q on the GPU decode path: the tables of the host path; the warning about reads without qualities; a file
    whose qualities are all above the threshold goes through the unmasked kernel and still gives the -Q 0 tables."""
    from mapdamage_amd import fasta
    from mapdamage_amd.main import main
    ref, b, rg, path = _write(tmp_path, n=20_000, seed=8)
    # a few records without qualities
    b2 = b
    rng = np.random.default_rng(3)
    for i in rng.choice(b2.n, 40, replace=False):
        b2.qual[b2.seq_off[i]:b2.seq_off[i + 1]] = 0xFF
    p2 = tmp_path / "noq.bam"
    sam.write_bam(str(p2), b2, ref.names, ref.lengths, RGS, rg_of_record=rg)
    fasta.write_fasta(tmp_path / "ref.fa", ref)
    outs = {}
    for name, flags in (("host", ["--host-decode"]), ("dev", ["--gpu-decode", "--chunk-mb", "2"])):
        out = tmp_path / name
        assert main(["-i", str(p2), "-r", str(tmp_path / "ref.fa"), "-d", str(out), "--no-stats", "-Q", "25"] + flags) == 0
        outs[name] = [(out / f).read_text() for f in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt")]
        log = (out / "Runtime_log.txt").read_text()
        assert log.count("Reads without PHRED scores found") == 1, name
    assert outs["host"] == outs["dev"]
    assert "decoding on the host" not in (tmp_path / "dev" / "Runtime_log.txt").read_text()
    # threshold below every quality: nothing to mask
    lo = int(b.qual[b.qual != 0xFF].min())
    res = {}
    for name, flags in (("q0", ["--host-decode"]), ("qlow", ["-Q", str(max(lo, 1)), "--gpu-decode"])):
        out = tmp_path / name
        assert main(["-i", str(path), "-r", str(tmp_path / "ref.fa"), "-d", str(out), "--no-stats"] + flags) == 0
        res[name] = [(out / f).read_text() for f in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt")]
    assert res["q0"] == res["qlow"]


def test_the_device_decoder_folds_the_mask_into_the_seq_column(tmp_path):
    """--min-basequal on the device path: the unpack kernel sees every quality byte and writes the SEQ column the packed
    masked kernel reads (MDX_SEQ_4BITQ, include/mdx.h; align.py:65-71: a base whose quality is below the threshold is stored
    as the complement of its code, 0xFF — no qualities — is not below any threshold) — no pass over the quality column in
    front of the launches.  Held against the quality column of the same view, slab by slab, with several libraries (the
    launches order batch and mask by library)."""
    from mapdamage_amd.engine import DamageEngine
    from tests.util import assert_tables_equal, oracle_tableset
    ref, b, rg, path = _write(tmp_path, n=25_000, seed=12)
    for i in np.random.default_rng(5).choice(b.n, 30, replace=False):
        b.qual[b.seq_off[i]:b.seq_off[i + 1]] = 0xFF
    sam.write_bam(str(path), b, ref.names, ref.lengths, RGS, rg_of_record=rg)
    libs = [("s", "lib1"), ("s", "lib2")]
    lib_of = {"rgA": 0, "rg_b2": 1, "x": 0}
    b.lib = np.array([lib_of[r] for r in rg], np.uint16)
    Q = 22
    want = oracle_tableset(ref, b, libs, 70, 10, Q)
    lut = np.zeros(256, np.uint8)
    lut[ord("A")], lut[ord("C")], lut[ord("T")], lut[ord("G")] = 1, 2, 4, 8
    at = 0
    with DamageEngine(libs, 70, 10, Q) as eng:
        eng.set_reference(ref)
        with sam.GpuBamStream(eng, str(path), readgroups=list(lib_of.items()), chunk_bytes=1 << 19, want_qual=True, min_basequal=Q) as g:
            assert g.packed
            slabs = 0
            while True:
                view = g.next_view()
                if view is None:
                    break
                assert view.qual and view.seq_format == 2 and not view.lowq
                nb = int(view.n_bases)
                qual = _d2h(view.qual, nb, np.uint8)
                nib = _d2h(view.seq, (nb + 1) // 2, np.uint8)
                codes = np.stack([nib & 15, nib >> 4], axis=1).reshape(-1)[:nb]
                plain = lut[b.seq[at:at + nb]]
                expect = np.where(qual < Q, plain ^ 15, plain)
                np.testing.assert_array_equal(codes, expect)
                at += nb
                eng.tabulate_view(view)
                slabs += 1
            got = eng.finish()
            assert slabs > 2 and eng.packed_launches() == slabs
        # (a decoder at another threshold than its context's would hand out columns masked for the wrong one: refused)
        with pytest.raises(ValueError, match="not the --min-basequal of the context"):
            sam.GpuBamStream(eng, str(path), readgroups=list(lib_of.items()), chunk_bytes=1 << 19, want_qual=True, min_basequal=Q + 1)
    assert_tables_equal(got, want)


def test_the_fallback_without_the_chunked_host_decoder_counts_the_file_again(tmp_path, monkeypatch):
    """--chunk-mb 0: the host path reads the file in one piece and cannot take it up in the middle — a device decode that
    gives up part of the way hands the WHOLE file to it (no resume position), and the tables are those of a host run."""
    from mapdamage_amd import fasta
    from mapdamage_amd.main import main
    ref, b, rg, path = _write(tmp_path, n=40_000)
    fasta.write_fasta(tmp_path / "ref.fa", ref)
    monkeypatch.setenv("MDX_GBAM_SLAB_BYTES", str(1 << 20))
    outs = {}
    for name, flags, fail in (("host", ["--host-decode", "--chunk-mb", "0"], None), ("resumed", ["--gpu-decode", "--chunk-mb", "0"], "2")):
        if fail is None:
            monkeypatch.delenv("MDX_GBAM_FAIL_AT", raising=False)
        else:
            monkeypatch.setenv("MDX_GBAM_FAIL_AT", fail)
        out = tmp_path / name
        assert main(["-i", str(path), "-r", str(tmp_path / "ref.fa"), "-d", str(out), "--no-stats", "--log-level", "DEBUG"] + flags) == 0
        outs[name] = [(out / f).read_text() for f in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt")]
    monkeypatch.delenv("MDX_GBAM_FAIL_AT", raising=False)
    assert outs["host"] == outs["resumed"]
    log = (tmp_path / "resumed" / "Runtime_log.txt").read_text()
    assert "WARNING GPU decode path gave up" in log and "the whole file again" in log


def test_downsample_to_a_fraction_on_the_device_path(tmp_path):
    """-n 0.3 --downsample-seed 7 (reader.py:134-146): the device path draws on the host from the flag column of every slab,
    with the run's one generator, and marks the records that leave — the tables of the host path, which are the tables of
    the records the reference's own _downsample_to_fraction keeps (the same stream of draws: tests/golden/downsample.npz
    pins the function both paths call), over several slabs; a fixed number of reads stays the host decoder's."""
    import random

    from mapdamage_amd import fasta
    from mapdamage_amd.main import main
    from mapdamage_amd.reader import BAMReader
    from tests.util import oracle_tableset
    ref, b, rg, path = _write(tmp_path, n=30_000, seed=21)
    fasta.write_fasta(tmp_path / "ref.fa", ref)
    lib_of = {"rgA": 0, "rg_b2": 1, "x": 0}
    b.lib = np.array([lib_of[r] for r in rg], np.uint16)
    kept = np.nonzero((b.flag & 0xF04) == 0)[0]
    rand = random.Random(7)
    chosen = np.array([i for i in kept if rand.random() < 0.3])
    want = oracle_tableset(ref, b.take(chosen), [("s", "lib1"), ("s", "lib2")], 70, 10, 0)
    outs = {}
    for name, flags in (("dev", ["--gpu-decode", "--chunk-mb", "2"]), ("host", ["--host-decode", "--chunk-mb", "1"])):
        out = tmp_path / name
        assert main(["-i", str(path), "-r", str(tmp_path / "ref.fa"), "-d", str(out), "--no-stats", "-n", "0.3", "--downsample-seed", "7",
                     "--log-level", "DEBUG"] + flags) == 0
        outs[name] = [(out / f).read_text() for f in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt")]
        assert outs[name] == [want.misincorporation_text(), want.dnacomp_text(), want.lgdistribution_text()], name
    log = (tmp_path / "dev" / "Runtime_log.txt").read_text()
    assert "Decode path: device; fallbacks from the device path: 0" in log
    out = tmp_path / "fixed"
    assert main(["-i", str(path), "-r", str(tmp_path / "ref.fa"), "-d", str(out), "--no-stats", "-n", "500", "--downsample-seed", "7",
                 "--log-level", "DEBUG", "--gpu-decode"]) == 0
    assert "Decode path: host decoder" in (out / "Runtime_log.txt").read_text()
    r = BAMReader(str(path), downsample_to=500, downsample_seed=7)
    (fixed,) = list(r.iter_batches())
    want500 = oracle_tableset(ref, fixed, [("s", "lib1"), ("s", "lib2")], 70, 10, 0)
    assert (out / "misincorporation.txt").read_text() == want500.misincorporation_text()


def test_flag_bit_15_of_a_file_is_not_the_kernels_hint(tmp_path):
    """MDX_FLAG_QUAL_ABOVE_MIN (0x8000) is a hint the library sets itself; a file whose FLAG field carries that bit
    (htslib does not reject it) must still be masked by its qualities alone (align.py:53-73): every decoder clears it."""
    from mapdamage_amd import fasta
    from mapdamage_amd.main import main
    from tests.util import oracle_tableset
    ref, b, rg, _ = _write(tmp_path, n=8_000, seed=21)
    b.qual[:] = np.random.default_rng(1).integers(2, 20, size=b.qual.shape[0]).astype(np.uint8)   # everything maskable
    clean = b.flag.copy()
    # (... and 0x4000, MDX_FLAG_HAS_QUAL, the second hint)
    b.flag = (b.flag | np.uint16(0xC000)).astype(np.uint16)
    path = tmp_path / "bit15.bam"
    sam.write_bam(str(path), b, ref.names, ref.lengths, RGS, rg_of_record=rg)
    assert (sam.read_bam(str(path)).batch.flag == clean).all()
    assert (sam.read_bam_native(str(path)).batch.flag == clean).all()
    fasta.write_fasta(tmp_path / "ref.fa", ref)
    outs = {}
    for name, flags in (("host", ["--host-decode"]), ("dev", ["--gpu-decode"])):
        out = tmp_path / name
        assert main(["-i", str(path), "-r", str(tmp_path / "ref.fa"), "-d", str(out), "--no-stats", "-Q", "25"] + flags) == 0
        outs[name] = [(out / f).read_text() for f in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt")]
    assert outs["host"] == outs["dev"]
    assert "decoding on the host" not in (tmp_path / "dev" / "Runtime_log.txt").read_text()
    # the oracle masks by the qualities alone
    libs = [("s", "lib1"), ("s", "lib2")]
    b.flag = clean
    b.lib = np.asarray([0 if r in ("rgA", "x") else 1 for r in rg], np.uint16)
    want = oracle_tableset(ref, b, libs, 70, 10, 25)
    assert outs["host"][0] == want.misincorporation_text() and outs["host"][1] == want.dnacomp_text()


def _inflate_on_device(eng, cases):
    """cases: [(deflate payload, ISIZE, CRC32)] -> [(status, bytes out, crc ok)] through mdx_gbam_inflate_blocks."""
    lib = eng._lib
    lib.mdx_gbam_inflate_blocks.restype = ctypes.c_int
    lib.mdx_gbam_inflate_blocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32,
                                            ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    comp = np.frombuffer(b"".join(c for c, _, _ in cases) + b"\0" * 8, np.uint8).copy()
    blk = np.zeros((len(cases), 4), np.uint32)
    cin = cout = 0
    for i, (c, isize, _) in enumerate(cases):
        blk[i] = (cin, len(c), cout, isize)
        cin += len(c)
        cout += (isize + 15) & ~15
    out = np.zeros(cout + 16, np.uint8)
    status = np.zeros(len(cases), np.int32)
    crc = np.asarray([k for _, _, k in cases], np.uint32)
    crc_ok = np.zeros(len(cases), np.uint8)
    rc = lib.mdx_gbam_inflate_blocks(eng._ctx, comp.ctypes.data, cin, blk.ctypes.data, len(cases), out.ctypes.data, cout,
                                     status.ctypes.data, crc.ctypes.data, crc_ok.ctypes.data)
    assert rc == 0, rc
    return [(int(status[i]), out[int(blk[i, 2]):int(blk[i, 2]) + int(blk[i, 3])].tobytes(), bool(crc_ok[i])) for i in range(len(cases))]


def test_device_inflate_against_zlib_every_block_type_and_damage():
    """gbam_inflate_kernel on streams htslib never writes but DEFLATE allows: levels 0/1/6/9 x default / Z_FIXED /
    Z_HUFFMAN_ONLY / Z_RLE, several deflate blocks per member (stored ones among them), matches at distances that
    cross the 4 KiB LDS ring and reach the output already flushed to HBM (4097 .. 32768), 64 KiB of one byte; then
    damaged streams: whatever zlib decodes the device decodes to the same bytes, what zlib refuses is refused or
    caught by the CRC — never accepted with other bytes, never a hang."""
    import random
    import zlib
    from mapdamage_amd.engine import DamageEngine

    def raw(data, level, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        if not flush_every:
            return c.compress(data) + c.flush()
        out = b""
        for lo in range(0, len(data), flush_every):
            out += c.compress(data[lo:lo + flush_every]) + c.flush(zlib.Z_FULL_FLUSH if (lo // flush_every) % 2 else zlib.Z_SYNC_FLUSH)
        return out + c.flush()

    rnd = random.Random(5)
    nprng = np.random.default_rng(5)
    rb = lambda n: nprng.integers(0, 256, n, dtype=np.uint8).tobytes()      # noqa: E731
    dna = lambda n: bytes(nprng.choice(np.frombuffer(b"ACGT", np.uint8), n))   # noqa: E731
    datas = [b"", b"a", b"abc" * 1000, rb(60000), dna(65536), b"\x00" * 65536, b"G" * 65536, rb(100),
             bytes(nprng.choice(np.frombuffer(b"ACGTN!#$%&'()*+,-./0123456789", np.uint8), 30000)),
             b"ACGT" * 3000 + rb(30000) + b"TTTTGGGG" * 2000, rb(20000) + dna(40000)]
    # matches that reach back across the ring into flushed output: a seed repeated at chosen distances
    for dist in (4095, 4096, 4097, 4098, 4350, 6143, 8191, 8192, 8193, 12000, 16384, 20000, 32767, 32768):
        seed = rb(700)
        body = bytearray(rb(dist - 700) if dist > 700 else b"")
        data = seed + bytes(body) + seed + dna(300) + seed[:258] + rb(50)
        datas.append(data[:65536])
        # the same right behind a stored block (level 0 first half, compressed second half in one member)
        datas.append((rb(dist) + data)[:65536])
    valid = []
    for d in datas:
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                valid.append((raw(d, level, strategy), d))
        for every in (1000, 5000, 33000):
            valid.append((raw(d, 6, flush_every=every), d))
            valid.append((raw(d, 9, zlib.Z_FIXED, flush_every=every), d))
    # a stored block right in front of a far match (distance > ring directly behind a stored block)
    for dist in (4097, 9000, 32768):
        a = rb(dist)
        c = zlib.compressobj(0, zlib.DEFLATED, -15)
        part1 = c.compress(a) + c.flush(zlib.Z_FULL_FLUSH)
        d2 = a[:300] + b"xyz"
        # second part compressed with the first as dictionary-less history: decoders see one stream
        c2 = zlib.compressobj(9, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY, a[-32768:])
        part2 = c2.compress(d2) + c2.flush()
        valid.append((part1 + part2, a + d2))

    def zlib_says(comp):
        try:
            z = zlib.decompressobj(-15)
            out = z.decompress(comp, 65537)
            if z.eof and len(out) <= 65536:
                return out
        except zlib.error:
            pass
        return None

    cases, expect = [], []
    for comp, d in valid:
        assert zlib_says(comp) == d
        cases.append((comp, len(d), zlib.crc32(d) & 0xFFFFFFFF))
        expect.append(("valid", d))
    for _ in range(200):                                   # random bytes
        g = rb(rnd.randint(1, 400))
        z = zlib_says(g)
        ref = z if z is not None else b"?" * rnd.randint(1, 500)
        cases.append((g, len(ref), zlib.crc32(ref) & 0xFFFFFFFF))
        expect.append(("ok" if z is not None else "refused", ref))
    for _ in range(1500):                                  # valid streams with a few bits flipped
        d = bytes(rnd.choice(b"ACGTACGTACGTNIIIIII#####") for _ in range(rnd.randint(1, 3000))) if rnd.random() < 0.7 else \
            (rb(rnd.randint(1, 9000)) + dna(rnd.randint(1, 9000))) * rnd.randint(1, 3)
        d = d[:65536]
        comp = bytearray(raw(d, rnd.choice([1, 6, 9]), rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE])))
        for _ in range(rnd.randint(1, 3)):
            comp[rnd.randrange(len(comp))] ^= 1 << rnd.randrange(8)
        comp = bytes(comp)
        if rnd.random() < 0.2:
            comp = comp[:rnd.randrange(1, len(comp) + 1)]                      # truncated
        z = zlib_says(comp)
        ref = z if z is not None else d
        cases.append((comp, len(ref), zlib.crc32(ref) & 0xFFFFFFFF))
        expect.append(("ok" if z is not None else "refused", ref))
    with DamageEngine([("s", "l")]) as eng:
        got = []
        for lo in range(0, len(cases), 512):
            got += _inflate_on_device(eng, cases[lo:lo + 512])
    bad = []
    for i, ((kind, ref), (status, out, crc_ok)) in enumerate(zip(expect, got)):
        if kind in ("valid", "ok"):
            if status != len(ref) or out != ref or not crc_ok:
                bad.append((i, kind, status, len(ref), crc_ok))
        elif status == len(ref) and crc_ok and out != ref:
            bad.append((i, kind, status, len(ref), crc_ok))
    assert not bad, bad[:10]
    assert sum(1 for k, _ in expect if k == "refused") > 300 and sum(1 for k, _ in expect if k == "ok") > 20


def test_device_bgzf_writer_is_read_by_every_decoder(tmp_path):
    """mdx_bgzf_deflate (the rescaling pass's writer, rescale.py:290-291, :344): a stream of encoded records cut into blocks of
    0xFF00 bytes, a lane per block on the device — the members must inflate (Python's gzip: header, CRC32, ISIZE checked) to the
    bytes that went in, keep BGZF's BSIZE field, and, written as a BAM file, read back record for record through the host
    decoder and the device decoder of this library."""
    import gzip
    import struct

    from mapdamage_amd.engine import DamageEngine
    ref, b, rg, _ = _write(tmp_path, n=30_000, seed=31)
    plain = tmp_path / "plain.bam"
    sam.write_bam(str(plain), b, ref.names, ref.lengths, RGS, rg_of_record=rg)
    stream = gzip.decompress(plain.read_bytes())              # header + records, as htslib would hand them to bgzf_write
    rnd = np.random.default_rng(3)
    with DamageEngine([("s", "lib1"), ("s", "lib2")]) as eng:
        for data in (stream, b"", b"x", stream[:0xFF00], stream[:0xFF00 + 1], rnd.integers(0, 256, 200_000, dtype=np.uint8).tobytes(),
                     b"\x00" * 300_000):
            members = bytes(eng.bgzf_deflate(data))
            assert gzip.decompress(members) == data if data else members == b""
            at, n_members = 0, 0
            while at < len(members):
                assert members[at:at + 4] == b"\x1f\x8b\x08\x04" and members[at + 12:at + 16] == b"BC\x02\x00"
                at += struct.unpack_from("<H", members, at + 16)[0] + 1
                n_members += 1
            assert at == len(members) and n_members == (len(data) + 0xFF00 - 1) // 0xFF00
        out = tmp_path / "device.bam"
        out.write_bytes(bytes(eng.bgzf_deflate(stream)) + sam._bgzf_block(b""))
        assert out.stat().st_size < 1.1 * plain.stat().st_size            # (zlib level 6 wrote the other one)
        back = sam.read_bam_native(str(out))
        for k in ("flag", "tid", "pos", "tlen", "cigar", "seq", "qual"):
            np.testing.assert_array_equal(getattr(back.batch, k), getattr(b, k), err_msg=k)
        eng.set_reference(ref)
        lib_of = {"rgA": 0, "rg_b2": 1, "x": 0}
        with sam.GpuBamStream(eng, str(out), readgroups=list(lib_of.items()), chunk_bytes=1 << 20) as g:
            n = 0
            while True:
                v = g.next_view()
                if v is None:
                    break
                n += int(v.n_reads)
        assert n == b.n
