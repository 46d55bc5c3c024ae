"""Decoders against BAM files this repository's writer did not produce: tests/golden/foreign*.bam, assembled byte by byte
from the SAM/BAM specification by tools/make_foreign_bam.py (multi-RG header, aux fields of every type in front of and
behind RG, B arrays, = / X / N / P operations, a 66 000-operation CIGAR in CG:B,I, 0xFF qualities, a secondary record
without SEQ, an unmapped record, records straddling unevenly cut BGZF blocks, no EOF marker), with the truth written down
at construction time (foreign_bam.npz).  What the reference sees of such a file through pysam: SURVEY Appendix C;
reader.py:63-81,99-118 for the read groups."""
import pathlib

import numpy as np
import pytest

from mapdamage_amd import sam
from mapdamage_amd.batch import ReadBatch, Reference

GOLDEN = pathlib.Path(__file__).resolve().parent / "golden"
COLS = ("flag", "tid", "pos", "tlen", "cigar_off", "cigar", "seq_off")
LIBS = [("sampleX", "libA"), ("sampleX", "libB"), ("sampleY", "libA")]      # header order of first appearance
LIB_OF_RG = {"lane1": 0, "lane.2": 1, "L3": 2}


def truth(tag):
    z = np.load(GOLDEN / "foreign_bam.npz")
    return {k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + "_")}, z


def reference(z):
    return Reference(["chrA", "chrB"], [bytes(z["genome_chrA"]), bytes(z["genome_chrB"])])


def check_alignments(al, t):
    b = al.batch
    for k in COLS:
        np.testing.assert_array_equal(getattr(b, k), t[k], err_msg=k)
    n = int(t["seq_off"][-1])
    np.testing.assert_array_equal(b.seq[:n], t["seq"])
    np.testing.assert_array_equal(b.qual[:n], t["qual"])
    np.testing.assert_array_equal(b.mtid, t["mtid"])
    np.testing.assert_array_equal(b.mpos, t["mpos"])
    assert [al.qname_at(i) for i in range(b.n)] == [str(x) for x in t["name"]]
    assert [al.rg[i] for i in range(b.n)] == [str(r) if h else None for r, h in zip(t["rg"], t["has_rg"])]


@pytest.mark.parametrize("tag,name", [("full", "foreign.bam"), ("nocg", "foreign_nocg.bam")])
def test_host_decoders_against_construction_time_truth(tag, name):
    t, z = truth(tag)
    path = str(GOLDEN / name)
    check_alignments(sam.read_bam_native(path), t)          # csrc/mdx_bamio.cpp, one piece
    check_alignments(sam.read_bam(path), t)                 # the Python cross-check
    # ... and in chunks of a few kilobytes of records (the 320 KB record is a chunk of its own)
    got = {k: [] for k in ("flag", "tid", "pos", "tlen", "nc", "ns")}
    seqs, cigs = [], []
    with sam.BamStream(path, chunk_bytes=3000) as stream:
        while True:
            chunk = stream.next_chunk()
            if chunk is None:
                break
            b = chunk.batch
            for k in ("flag", "tid", "pos", "tlen"):
                got[k].append(getattr(b, k))
            got["nc"].append(np.diff(b.cigar_off.astype(np.int64)))
            got["ns"].append(np.diff(b.seq_off.astype(np.int64)))
            seqs.append(b.seq[:int(b.seq_off[-1])])
            cigs.append(b.cigar[:int(b.cigar_off[-1])])
    for k in ("flag", "tid", "pos", "tlen"):
        np.testing.assert_array_equal(np.concatenate(got[k]), t[k])
    np.testing.assert_array_equal(np.concatenate(got["nc"]), np.diff(t["cigar_off"].astype(np.int64)))
    np.testing.assert_array_equal(np.concatenate(got["ns"]), np.diff(t["seq_off"].astype(np.int64)))
    np.testing.assert_array_equal(np.concatenate(seqs), t["seq"])
    np.testing.assert_array_equal(np.concatenate(cigs), t["cigar"])


def test_the_script_rebuilds_the_committed_truth():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_foreign_bam", str(GOLDEN.parent.parent / "tools" / "make_foreign_bam.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _, records, genome, _ = mod.build()
    t, z = truth("full")
    arrays = mod.truth_arrays(records)
    for k in COLS + ("seq", "qual"):
        np.testing.assert_array_equal(arrays[k], t[k])
    np.testing.assert_array_equal(genome[0], z["genome_chrA"])
    # the long record: its CIGAR field in the file is the placeholder, its operations live in the CG tag
    i = [str(x) for x in t["name"]].index("long_cigar")
    assert int(t["cigar_off"][i + 1] - t["cigar_off"][i]) == 66_000


def test_read_groups_of_the_foreign_header():
    """reader.py:99-118: (SM, LB) per @RG line whatever the order of its tags and whatever else it carries; a line
    without LB is the reference's BAMError unless --merge-libraries."""
    from mapdamage_amd.reader import BAMReader
    from mapdamage_amd.sam import BAMError
    t, _ = truth("nocg")
    r = BAMReader(str(GOLDEN / "foreign_nocg.bam"))
    assert r.get_libraries() == LIBS
    (batch,) = list(r.iter_batches())
    kept = np.nonzero((t["flag"] & 0xF04) == 0)[0]
    np.testing.assert_array_equal(batch.pos, t["pos"][kept])
    np.testing.assert_array_equal(batch.lib, [LIB_OF_RG[str(t["rg"][i])] for i in kept])
    with pytest.raises(BAMError, match="Incomplete readgroup found: nolb is missing 'LB'"):
        BAMReader(str(GOLDEN / "foreign_nolb.bam"))
    merged = BAMReader(str(GOLDEN / "foreign_nolb.bam"), merge_libraries=True)
    assert merged.get_libraries() == [("*", "*")]
    assert sum(b.n for b in merged.iter_batches()) == len(kept)


def _truth_batch(t):
    b = ReadBatch(t["flag"].copy(), np.zeros(len(t["flag"]), np.uint16), t["tid"].copy(), t["pos"].copy(), t["tlen"].copy(), t["cigar_off"].copy(),
                  t["cigar"].copy(), t["seq_off"].copy(), t["seq"].copy(), t["qual"].copy())
    b.lib = np.array([LIB_OF_RG.get(str(r), 0) for r in t["rg"]], np.uint16)
    return b


@pytest.mark.gpu
def test_device_decoder_against_construction_time_truth():
    """csrc/mdx_gbam.hip on the file without the CG record: columns, bases in both forms of the SEQ column, qualities,
    libraries (0xFFFF: no RG tag); the file WITH the CG record is MDX_ERR_UNSUPPORTED on the device path."""
    from mapdamage_amd.engine import DamageEngine
    from tests.test_gpu_decode import _d2h
    t, z = truth("nocg")
    ref = reference(z)
    n = len(t["flag"])
    for packed in (False, True):
        with DamageEngine(LIBS) as eng:
            eng.set_reference(ref)
            with sam.GpuBamStream(eng, str(GOLDEN / "foreign_nocg.bam"), readgroups=list(LIB_OF_RG.items()), chunk_bytes=1 << 16,
                                  want_qual=True, want_mate=True, packed=packed) as g:
                view = g.next_view()
                assert int(view.n_reads) == n and g.next_view() is None
                for k, dt in (("tid", np.int32), ("pos", np.int32), ("tlen", np.int32)):
                    np.testing.assert_array_equal(_d2h(getattr(view, k), n, dt), t[k], err_msg=k)
                # the file's flag bits, and on top the hint MDX_FLAG_HAS_QUAL (0x4000) where the record has qualities: a base
                # at least, and a first quality byte that is not 0xFF (main.py:185, rescale.py:306)
                flag = _d2h(view.flag, n, np.uint16)
                np.testing.assert_array_equal(flag & np.uint16(0x3FFF), t["flag"], err_msg="flag")
                lens = np.diff(t["seq_off"].astype(np.int64))
                first = t["qual"][np.minimum(t["seq_off"][:-1].astype(np.int64), len(t["qual"]) - 1)]
                np.testing.assert_array_equal((flag & 0x4000) != 0, (lens > 0) & (first != 0xFF))
                np.testing.assert_array_equal(_d2h(view.cigar_off, n + 1, np.uint32), t["cigar_off"])
                np.testing.assert_array_equal(_d2h(view.seq_off, n + 1, np.uint32), t["seq_off"])
                np.testing.assert_array_equal(_d2h(view.cigar, int(view.n_cigar), np.uint32), t["cigar"])
                nb = int(view.n_bases)
                np.testing.assert_array_equal(_d2h(view.qual, nb, np.uint8), t["qual"])
                np.testing.assert_array_equal(_d2h(view.mtid, n, np.int32), t["mtid"])
                np.testing.assert_array_equal(_d2h(view.mpos, n, np.int32), t["mpos"])
                lib = _d2h(view.lib, n, np.uint16)
                np.testing.assert_array_equal(lib, [LIB_OF_RG[str(r)] if h else 0xFFFF for r, h in zip(t["rg"], t["has_rg"])])
                if packed:
                    nib = _d2h(view.seq, (nb + 1) // 2, np.uint8)
                    codes = np.stack([nib & 15, nib >> 4], axis=1).reshape(-1)[:nb]
                    lut = np.zeros(256, np.uint8)
                    lut[ord("A")], lut[ord("C")], lut[ord("T")], lut[ord("G")] = 1, 2, 4, 8
                    np.testing.assert_array_equal(codes, lut[t["seq"]])
                else:
                    np.testing.assert_array_equal(_d2h(view.seq, nb, np.uint8), t["seq"])
    with DamageEngine(LIBS) as eng:
        eng.set_reference(ref)
        with pytest.raises(sam.GpuDecodeUnsupported, match="CG tag"):
            with sam.GpuBamStream(eng, str(GOLDEN / "foreign.bam"), readgroups=list(LIB_OF_RG.items()), chunk_bytes=1 << 16) as g:
                while g.next_view() is not None:
                    pass


@pytest.mark.gpu
@pytest.mark.parametrize("name,tag", [("foreign.bam", "full"), ("foreign_nocg.bam", "nocg")])
@pytest.mark.parametrize("extra", [[], ["-Q", "25"], ["--merge-libraries"]])
def test_cli_tables_of_the_foreign_files(tmp_path, name, tag, extra):
    """`python -m mapdamage_amd` on the hand-assembled files (the device path where it applies, the host decoder behind the CG
    record): the tables of the oracle over the construction-time records, per library, with --min-basequal, merged."""
    from mapdamage_amd import fasta
    from mapdamage_amd.main import main
    from tests.util import oracle_tableset
    t, z = truth(tag)
    ref = reference(z)
    fasta.write_fasta(tmp_path / "ref.fa", ref)
    b = _truth_batch(t)
    libs = LIBS
    if "--merge-libraries" in extra:
        b.lib[:] = 0
        libs = [("*", "*")]
    Q = 25 if "-Q" in extra else 0
    want = oracle_tableset(ref, b, libs, 70, 10, Q)
    texts = {}
    for decode in ("--gpu-decode", "--host-decode"):
        out = tmp_path / decode.strip("-")
        assert main(["-i", str(GOLDEN / name), "-r", str(tmp_path / "ref.fa"), "-d", str(out), "--no-stats", decode] + extra) == 0
        texts[decode] = [(out / f).read_text() for f in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt")]
        assert texts[decode] == [want.misincorporation_text(), want.dnacomp_text(), want.lgdistribution_text()], decode
    log = (tmp_path / "gpu-decode" / "Runtime_log.txt").read_text()
    assert ("GPU decode path gave up" in log) == (tag == "full")


@pytest.mark.gpu
def test_rescaling_a_foreign_file_on_the_device_writes_what_the_host_route_writes(tmp_path):
    """--rescale-only over records this repository's writer did not produce (aux fields of every type in front of and behind RG,
    B arrays, records straddling unevenly cut BGZF blocks, 0xFF qualities, an unmapped record, a record without SEQ): the route
    that never brings the records to the host (mdx_gbam_rescale_slab: the new qualities patched into the inflated records, the
    tags walked for an MR, MR:f appended, the stream compressed on the device) must write the records the host route writes —
    every encoded byte of every record — and both must hold the oracle's qualities.  The file is foreign_nocg.bam less the
    records the reference's rescaling cannot take (the oracle says which: rescale.py raises on them), their encoded bytes
    carried over as they stand."""
    from mapdamage_amd.engine import BadReadError, DamageEngine
    from mapdamage_amd.rescale import RescaleModel, rescale_bam, rescale_bam_on_device
    from oracle import oracle
    t, z = truth("nocg")
    ref = reference(z)
    rng = np.random.default_rng(5)
    corr_prob = {}
    for p in list(range(1, 13)) + list(range(-12, 0)):
        corr_prob[("C", "T", p)] = float(rng.random() * 0.8)
        corr_prob[("G", "A", p)] = float(rng.random() * 0.8)
    model = RescaleModel(corr_prob, 12, 12)
    corr = np.zeros((2, model.npos))
    for (r_, _s, p_), v in corr_prob.items():
        corr[0 if r_ == "C" else 1, p_ if p_ > 0 else model.len5p - p_] = v
    al = sam.read_bam(GOLDEN / "foreign_nocg.bam", keep_raw=True)
    whole = al.batch
    keep = list(range(whole.n))
    while True:
        b = whole.take(np.asarray(keep))
        try:
            wq, wmr, wst = oracle.rescale(ref, b, corr, 12, 12)
            break
        except oracle.OracleError as error:
            keep.pop(int(error.read_index))
    assert 0 < len(keep) < whole.n                       # (some records of the file are not the rescaling's to take, most are)
    src = tmp_path / "foreign_rescalable.bam"
    sam.write_bam_raw(src, al.raw_header, [bytes(al.raw[i]) for i in keep])
    # (the whole file stops every route at the same record)
    for fn in (rescale_bam_on_device, rescale_bam):
        with DamageEngine([("*", "*")]) as eng:
            with pytest.raises(BadReadError):
                fn(eng, ref, GOLDEN / "foreign_nocg.bam", tmp_path / "never.bam", model)
    outs = {}
    for name, fn, kw in (("device", rescale_bam_on_device, dict(slab_bytes=1 << 16)), ("host", rescale_bam, dict(chunk_bytes=3000, device_deflate=False)),
                         ("host_device_deflate", rescale_bam, dict(chunk_bytes=1 << 20))):
        with DamageEngine([("*", "*")]) as eng:
            _, counts = fn(eng, ref, src, tmp_path / (name + ".bam"), model, **kw)
        outs[name] = (counts, sam.read_bam(tmp_path / (name + ".bam"), keep_raw=True))
    counts, back = outs["device"]
    assert back.batch.n == len(keep) and sum(counts.values()) == back.batch.n
    for other in ("host", "host_device_deflate"):
        assert outs[other][0] == counts
        assert [bytes(x) for x in outs[other][1].raw] == [bytes(x) for x in back.raw]
    # ... and the qualities are the oracle's, the tag where it rescaled
    np.testing.assert_array_equal(back.batch.qual, wq)
    np.testing.assert_array_equal(np.asarray(back.has_mr), ~np.isnan(wmr))
    for k in ("flag", "tid", "pos", "tlen", "cigar", "seq", "mtid", "mpos"):
        np.testing.assert_array_equal(getattr(back.batch, k), getattr(b, k), err_msg=k)
