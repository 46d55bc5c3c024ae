"""Quality rescaling (mapdamage/rescale.py; BASELINE configs[4]).  The golden holds the output of the
reference's own _rescale_qual_core over a stand-in AlignmentFile (tools/ref_harness.py)."""

import json
import pathlib

import numpy as np
import pytest

from mapdamage_amd import synth
from mapdamage_amd.batch import ReadBatch, Reference
from mapdamage_amd.rescale import RescaleModel, finalize_mr, get_corr_prob

GOLDEN = pathlib.Path(__file__).resolve().parent / "golden" / "genome_rescale.npz"


def load(tmp_path):
    z = np.load(GOLDEN)
    names = json.loads(bytes(z["names"]).decode())
    seqs, o = [], 0
    bases = bytes(z["ref_bases"])
    for ln in z["ref_lengths"]:
        seqs.append(bases[o:o + int(ln)])
        o += int(ln)
    batch = ReadBatch(z["flag"], np.zeros(len(z["flag"]), np.uint16), z["tid"], z["pos"], z["tlen"], z["cigar_off"],
                      z["cigar"], z["seq_off"], z["seq"], z["qual"], z["mtid"], z["mpos"]).validate()
    path = tmp_path / "Stats_out_MCMC_correct_prob.csv"
    path.write_bytes(bytes(z["csv"]))
    len5p, len3p = int(z["len5p"]), int(z["len3p"])
    model = RescaleModel.from_csv(path, len5p, len3p)
    return Reference(names, seqs), batch, model, get_corr_prob(path, len5p, len3p), z["qual_out"], z["mr"]


def golden_summary_lines():
    """The reference's own _print_subs output (rescale.py:159-192) captured by tools/ref_harness.py."""
    log = json.loads(bytes(np.load(GOLDEN)["log"]).decode())
    start = log.index("Expected substition frequencies before and after rescaling:")
    return log[start:]


def corr_table(corr_prob, model):
    corr = np.zeros((2, model.npos))
    for (r, _s, p), v in corr_prob.items():
        corr[0 if r == "C" else 1, p if p > 0 else model.len5p - p] = v
    return corr


def check(qual_out, mr_raw, want_qual, want_mr):
    np.testing.assert_array_equal(qual_out, want_qual)
    assert np.array_equal(np.isnan(mr_raw), np.isnan(want_mr))
    got = np.asarray([np.nan if np.isnan(m) else finalize_mr(m) for m in mr_raw])
    np.testing.assert_array_equal(got[~np.isnan(got)], want_mr[~np.isnan(want_mr)])


def test_mr_rounding_in_the_library_is_pythons():
    """``float("%.5f" % x)`` (rescale.py:275-276) for every MR sum of a chunk at once (include/mdx.h mdx_mr_round) against the
    expression itself, ties and tiny values included."""
    from mapdamage_amd.rescale import finalize_mr, round_mr
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.random(50_000) * 3, rng.random(1000) * 1e-5, np.arange(0, 2000) * 5e-6 + 5e-6,
                        [np.nan, 0.0, 0.000005, 0.000015, 0.123455, 2.5e-6, 1e10 / 3, 1e-300]])
    want = np.array([0.0 if v != v else np.float32(finalize_mr(v)) for v in x], np.float32)
    np.testing.assert_array_equal(round_mr(x), want)


def test_lookup_table_matches_survey_probe():
    """SURVEY Appendix D rescale probe: q = 40, corr 0 / 0.01 / 0.1 / 0.25 / 0.5 / 0.9."""
    cp = {("C", "T", 1): 0.5, ("C", "T", 2): 0.25, ("C", "T", 3): 0.1, ("C", "T", 4): 0.01, ("G", "A", -1): 0.9}
    m = RescaleModel(cp, 12, 12)
    assert [int(m.lut[0, k, 40]) for k in (0, 4, 3, 2, 1)] == [40, 20, 10, 6, 3]
    assert int(m.lut[1, 13, 40]) == 0 and m.term[0, 1] == 0.5 and m.term[0, 0] == 0.0
    assert all(int(m.lut[0, 0, q]) == q for q in range(94))   # no correction: quality unchanged


def test_oracle_matches_reference_golden(tmp_path):
    from oracle import oracle
    ref, batch, model, corr_prob, want_qual, want_mr = load(tmp_path)
    qual_out, mr_raw, status = oracle.rescale(ref, batch, corr_table(corr_prob, model), model.len5p, model.len3p)
    check(qual_out, mr_raw, want_qual, want_mr)
    assert set(np.unique(status)) == {0, 1, 2, 3, 4}
    assert int((want_qual != batch.qual).sum()) > 100       # the fixture does rescale something


def test_oracle_summary_matches_reference_log(tmp_path):
    """R3: the substitution summary the reference logs after rescaling."""
    from oracle import oracle
    ref, batch, model, corr_prob, _, _ = load(tmp_path)
    _, _, _, counts, pvals = oracle.rescale_with_subs(ref, batch, corr_table(corr_prob, model), model.len5p, model.len3p)
    assert oracle.subs_log_lines(counts, pvals) == golden_summary_lines()
    assert len(golden_summary_lines()) == 16


def summary_ints_from_oracle(counts):
    """Oracle counts (130 qualities) -> the device layout's first 756 words (94 qualities)."""
    hist = counts[4:].reshape(4, 2, 130)
    assert int(hist[:, :, 94:].sum()) == 0
    return np.concatenate([counts[:4], hist[:, :, :94].reshape(-1)])


@pytest.mark.gpu
def test_hip_matches_reference_golden(tmp_path):
    from mapdamage_amd.engine import DamageEngine
    from mapdamage_amd.rescale import RescaleSummary
    ref, batch, model, corr_prob, want_qual, want_mr = load(tmp_path)
    with DamageEngine([("s", "l")]) as eng:
        eng.set_reference(ref)
        eng.set_rescale_model(model)
        qual_out, mr_raw, status = eng.rescale(batch)
        words = eng.rescale_summary()
    check(qual_out, mr_raw, want_qual, want_mr)
    assert RescaleSummary(words, model).log_lines() == golden_summary_lines()    # R3, against the reference's log


@pytest.mark.gpu
def test_hip_matches_oracle_seeded(tmp_path):
    from mapdamage_amd.engine import DamageEngine
    from oracle import oracle
    _, _, model, corr_prob, _, _ = load(tmp_path)
    ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500,
                            lower_run=3000)
    b = synth.make_reads(ref, 60_000, 21, len_range=(25, 160), paired=True, frac_softclip=0.2, frac_ins=0.08,
                         frac_del=0.08, frac_skip=0.01, with_qual=True, frac_filtered=0.03)
    rng = np.random.default_rng(3)
    b.mtid = np.where(rng.random(b.n) < 0.9, b.tid, (b.tid + 1) % 2).astype(np.int32)
    b.mpos = (b.pos + rng.integers(-300, 300, size=b.n)).astype(np.int32)
    b.flag = np.where(rng.random(b.n) < 0.5, b.flag & 0xF14, b.flag).astype(np.uint16)
    want_q, want_mr, want_st, want_counts, want_pvals = oracle.rescale_with_subs(
        ref, b, corr_table(corr_prob, model), model.len5p, model.len3p)
    with DamageEngine([("s", "l")]) as eng:
        eng.set_reference(ref)
        eng.set_rescale_model(model)
        got_q, got_mr, got_st = eng.rescale(b)
        words = eng.rescale_summary()
    from mapdamage_amd.rescale import RescaleSummary
    np.testing.assert_array_equal(words[:756], summary_ints_from_oracle(want_counts))      # bit-exact integers
    assert int(words[756:].reshape(2, -1, 94).sum()) == int(want_counts[4:4 + 130].sum() + want_counts[4 + 4 * 130:4 + 5 * 130].sum())
    assert RescaleSummary(words, model).log_lines() == oracle.subs_log_lines(want_counts, want_pvals)
    np.testing.assert_array_equal(got_q, want_q)
    np.testing.assert_array_equal(got_st, want_st)
    assert np.array_equal(np.isnan(got_mr), np.isnan(want_mr))
    np.testing.assert_array_equal(got_mr[~np.isnan(got_mr)], want_mr[~np.isnan(want_mr)])   # bit-exact fp64 sums


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(3))
def test_hip_rescale_fuzzed_cigars_match_oracle(k, tmp_path):
    """The rescaling pass on the fuzzed CIGARs of tools/fuzz_vs_reference.py (hard clips only where the reference
    can write the record — not outside a soft clip, rescale.py:266-271): qualities, MR, status, summary."""
    from mapdamage_amd.batch import batch_from_records
    from mapdamage_amd.engine import DamageEngine
    from oracle import oracle
    from tools.fuzz_vs_reference import fuzz_records, rescale_writable
    _, _, model, corr_prob, _, _ = load(tmp_path)
    ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500,
                            lower_run=3000)
    recs = [r for r in fuzz_records(ref, 5000, 8100 + k, with_qual=True) if rescale_writable(r["cigar"])]
    b = batch_from_records(recs, with_qual=True)
    rng = np.random.default_rng(30 + k)
    b.mtid = np.where(rng.random(b.n) < 0.9, b.tid, (b.tid + 1) % 2).astype(np.int32)
    b.mpos = (b.pos + rng.integers(-300, 300, size=b.n)).astype(np.int32)
    want_q, want_mr, want_st, want_counts, want_pvals = oracle.rescale_with_subs(
        ref, b, corr_table(corr_prob, model), model.len5p, model.len3p)
    with DamageEngine([("s", "l")]) as eng:
        eng.set_reference(ref)
        eng.set_rescale_model(model)
        got_q, got_mr, got_st = eng.rescale(b)
        words = eng.rescale_summary()
    np.testing.assert_array_equal(words[:756], summary_ints_from_oracle(want_counts))
    np.testing.assert_array_equal(got_q, want_q)
    np.testing.assert_array_equal(got_st, want_st)
    assert np.array_equal(np.isnan(got_mr), np.isnan(want_mr))
    np.testing.assert_array_equal(got_mr[~np.isnan(got_mr)], want_mr[~np.isnan(want_mr)])


@pytest.mark.gpu
@pytest.mark.parametrize("l5,l3", [(0, 0), (1, 30), (30, 2), (16, 16), (17, 17), (60, 45), (130, 130)])
def test_hip_rescale_window_lengths_match_oracle(l5, l3):
    """Models of other lengths than the usual 12 + 12: the end windows of the kernel's fast path take one round (up to
    16 columns each), several (longer), or the model does not fit its LDS image at all (130 + 130: every record by
    the walk kernel).  Short reads make the two windows meet or overlap."""
    from mapdamage_amd.engine import DamageEngine
    from mapdamage_amd.rescale import RescaleModel
    from oracle import oracle
    rng = np.random.default_rng(100 * l5 + l3)
    corr_prob = {}
    for p in list(range(1, l5 + 1)) + list(range(-l3, 0)):
        # (a few exact zeros: columns inside the windows whose key adds nothing)
        corr_prob[("C", "T", p)] = 0.0 if rng.random() < 0.1 else float(rng.random() * 0.6)
        corr_prob[("G", "A", p)] = 0.0 if rng.random() < 0.1 else float(rng.random() * 0.6)
    model = RescaleModel(corr_prob, l5, l3)
    ref = synth.make_genome(seed=12, sizes=(("chr1", 200_000), ("chr2", 50_000)), n_run=300, lower_run=2000)
    b = synth.make_reads(ref, 30_000, 40 + l5, len_range=(8, 200), paired=True, frac_softclip=0.2, frac_ins=0.1,
                         frac_del=0.1, frac_skip=0.01, with_qual=True, frac_filtered=0.03)
    b.mtid = np.where(rng.random(b.n) < 0.9, b.tid, (b.tid + 1) % 2).astype(np.int32)
    b.mpos = (b.pos + rng.integers(-300, 300, size=b.n)).astype(np.int32)
    b.flag = np.where(rng.random(b.n) < 0.5, b.flag & 0xF14, b.flag).astype(np.uint16)
    want_q, want_mr, want_st, want_counts, want_pvals = oracle.rescale_with_subs(
        ref, b, corr_table(corr_prob, model), model.len5p, model.len3p)
    with DamageEngine([("s", "l")]) as eng:
        eng.set_reference(ref)
        eng.set_rescale_model(model)
        got_q, got_mr, got_st = eng.rescale(b)
        words = eng.rescale_summary()
    from mapdamage_amd.rescale import RescaleSummary
    np.testing.assert_array_equal(words[:756], summary_ints_from_oracle(want_counts))
    assert RescaleSummary(words, model).log_lines() == oracle.subs_log_lines(want_counts, want_pvals)
    np.testing.assert_array_equal(got_q, want_q)
    np.testing.assert_array_equal(got_st, want_st)
    assert np.array_equal(np.isnan(got_mr), np.isnan(want_mr))
    np.testing.assert_array_equal(got_mr[~np.isnan(got_mr)], want_mr[~np.isnan(want_mr)])


def short_records(ref, seed, n=6000):
    """Records of 1 .. 45 bases — [S] M [S] and [S] M (I | D) M [S] with runs from one base up, both strands, paired
    and single, damage-like substitutions at a high rate so that nearly every window holds candidates, qualities at
    both ends of the 0..93 range and bytes that are not bases."""
    rng = np.random.default_rng(seed)
    bases, offs = ref.concat()
    upper = bases & np.uint8(0xDF)
    lens = list(ref.lengths)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    recs = []
    for i in range(n):
        tid = int(rng.integers(0, 2))
        kind = int(rng.integers(0, 3))           # 0 plain, 1 insertion, 2 deletion
        m1 = int(rng.integers(1, 46))
        ops = [(int(rng.choice([0, 7, 8])), m1)]
        if kind:
            ops += [(kind, int(rng.integers(1, 6))), (0, int(rng.integers(1, 46)))]
        span = sum(ln for op, ln in ops if op != 1)
        pos = int(rng.integers(0, 6)) if i % 50 == 0 else int(rng.integers(0, lens[tid] - span))
        seq, r = [], offs[tid] + pos
        for op, ln in ops:
            if op == 1:
                seq.append(rng.choice(acgt, ln))
            elif op == 2:
                r += ln
            else:
                seq.append(upper[r:r + ln].copy()); r += ln
        body = np.concatenate(seq)
        u = rng.random(body.shape[0])
        body = np.where((body == ord("C")) & (u < 0.35), ord("T"), body)
        body = np.where((body == ord("G")) & (u > 0.65), ord("A"), body)
        body = np.where(rng.random(body.shape[0]) < 0.05, rng.choice(np.frombuffer(b"ACGTNE", np.uint8), body.shape[0]), body)
        sl, sr = (int(rng.integers(1, 12)) if rng.random() < 0.3 else 0 for _ in range(2))
        seq = np.concatenate([rng.choice(acgt, sl), body, rng.choice(acgt, sr)]).astype(np.uint8)
        cig = ([(4, sl)] if sl else []) + ops + ([(4, sr)] if sr else [])
        flag = int(rng.choice([0, 16]))
        if rng.random() < 0.5:
            flag |= 0x1 | int(rng.choice([0x40, 0x80])) | (0x20 if rng.random() < 0.5 else 0)
        qual = np.where(rng.random(seq.shape[0]) < 0.05, rng.choice([0, 1, 92, 93], seq.shape[0]),
                        rng.integers(2, 42, seq.shape[0])).astype(np.uint8)
        recs.append(dict(flag=flag, tid=tid, pos=pos, cigar=cig, seq=seq.tobytes().decode("latin-1"), qual=qual, lib=0, tlen=0))
    return recs


@pytest.mark.gpu
@pytest.mark.parametrize("l5,l3", [(12, 12), (3, 20), (25, 7)])
def test_hip_rescale_short_records_match_oracle(l5, l3):
    """Reads shorter than the end windows (the windows meet, overlap, or a run of a gapped read is shorter than a
    window and the record goes to the walk kernel), the first records of the batch (no eight bytes in front of them)."""
    from mapdamage_amd.batch import batch_from_records
    from mapdamage_amd.engine import DamageEngine
    from mapdamage_amd.rescale import RescaleModel
    from oracle import oracle
    rng = np.random.default_rng(7 * l5 + l3)
    corr_prob = {}
    for p in list(range(1, l5 + 1)) + list(range(-l3, 0)):
        corr_prob[("C", "T", p)] = float(rng.random() * 0.7)
        corr_prob[("G", "A", p)] = float(rng.random() * 0.7)
    model = RescaleModel(corr_prob, l5, l3)
    ref = synth.make_genome(seed=13, sizes=(("chr1", 100_000), ("chr2", 30_000)), n_run=200, lower_run=1000)
    b = batch_from_records(short_records(ref, 500 + l5), with_qual=True)
    b.mtid = np.where(rng.random(b.n) < 0.9, b.tid, (b.tid + 1) % 2).astype(np.int32)
    b.mpos = (b.pos + rng.integers(-100, 100, size=b.n)).astype(np.int32)
    want_q, want_mr, want_st, want_counts, want_pvals = oracle.rescale_with_subs(
        ref, b, corr_table(corr_prob, model), model.len5p, model.len3p)
    assert ((want_st == 2) | (want_st == 3)).sum() > b.n // 2 and (want_q != b.qual).sum() > b.n
    with DamageEngine([("s", "l")]) as eng:
        eng.set_reference(ref)
        eng.set_rescale_model(model)
        got_q, got_mr, got_st = eng.rescale(b)
        words = eng.rescale_summary()
    np.testing.assert_array_equal(words[:756], summary_ints_from_oracle(want_counts))
    np.testing.assert_array_equal(got_q, want_q)
    np.testing.assert_array_equal(got_st, want_st)
    assert np.array_equal(np.isnan(got_mr), np.isnan(want_mr))
    np.testing.assert_array_equal(got_mr[~np.isnan(got_mr)], want_mr[~np.isnan(want_mr)])


from tools.make_golden_hardclip import hardclip_records  # noqa: E402


HARDCLIP_GOLDEN = pathlib.Path(__file__).resolve().parent / "golden" / "genome_rescale_hardclip.npz"


def load_hardclip(tmp_path):
    z = np.load(HARDCLIP_GOLDEN)
    names = json.loads(bytes(z["names"]).decode())
    seqs, o = [], 0
    bases = bytes(z["ref_bases"])
    for ln in z["ref_lengths"]:
        seqs.append(bases[o:o + int(ln)])
        o += int(ln)
    batch = ReadBatch(z["flag"], np.zeros(len(z["flag"]), np.uint16), z["tid"], z["pos"], z["tlen"], z["cigar_off"],
                      z["cigar"], z["seq_off"], z["seq"], z["qual"], z["mtid"], z["mpos"]).validate()
    path = tmp_path / "hc.csv"
    path.write_bytes(bytes(z["csv"]))
    model = RescaleModel.from_csv(path, int(z["len5p"]), int(z["len3p"]))
    return Reference(names, seqs), batch, model, get_corr_prob(path, int(z["len5p"]), int(z["len3p"])), z["qual_out"], z["mr"]


def test_oracle_matches_reference_golden_with_hard_clips(tmp_path):
    """The reference's own _rescale_qual_core over 50M5H / 5H50M / 5H50M5H / clip + indel records
    (tools/make_golden_hardclip.py): every such record is rescaled and written (rescale.py:266-271)."""
    from oracle import oracle
    ref, batch, model, corr_prob, want_qual, want_mr = load_hardclip(tmp_path)
    qual_out, mr_raw, status = oracle.rescale(ref, batch, corr_table(corr_prob, model), model.len5p, model.len3p)
    check(qual_out, mr_raw, want_qual, want_mr)
    assert not np.isnan(want_mr).any() and int((want_qual != batch.qual).sum()) > 50


@pytest.mark.gpu
def test_hip_matches_reference_golden_with_hard_clips(tmp_path):
    from mapdamage_amd.engine import DamageEngine
    ref, batch, model, _corr_prob, want_qual, want_mr = load_hardclip(tmp_path)
    with DamageEngine([("s", "l")]) as eng:
        eng.set_reference(ref)
        eng.set_rescale_model(model)
        qual_out, mr_raw, _status = eng.rescale(batch)
    check(qual_out, mr_raw, want_qual, want_mr)


def test_oracle_rescales_hard_clipped_records(tmp_path):
    """a CIGAR ending in H is rescaled like any other (rescale.py:266-271 only re-attaches S clips)."""
    from mapdamage_amd.batch import batch_from_records
    from oracle import oracle
    _, _, model, corr_prob, _, _ = load(tmp_path)
    ref = synth.small_genome()
    b = batch_from_records(hardclip_records(ref), with_qual=True)
    b.mtid = b.tid.copy(); b.mpos = b.pos.copy()
    q, mr, st = oracle.rescale(ref, b, corr_table(corr_prob, model), model.len5p, model.len3p)
    assert set(np.unique(st)) == {2} and not np.isnan(mr).any()
    assert int((q != b.qual).sum()) > 50


@pytest.mark.gpu
def test_hip_rescales_hard_clipped_records(tmp_path):
    from mapdamage_amd.batch import batch_from_records
    from mapdamage_amd.engine import DamageEngine
    from oracle import oracle
    _, _, model, corr_prob, _, _ = load(tmp_path)
    ref = synth.small_genome()
    b = batch_from_records(hardclip_records(ref), with_qual=True)
    b.mtid = b.tid.copy(); b.mpos = b.pos.copy()
    want_q, want_mr, want_st = oracle.rescale(ref, b, corr_table(corr_prob, model), model.len5p, model.len3p)
    with DamageEngine([("s", "l")]) as eng:
        eng.set_reference(ref)
        eng.set_rescale_model(model)
        got_q, got_mr, got_st = eng.rescale(b)
    np.testing.assert_array_equal(got_q, want_q)
    np.testing.assert_array_equal(got_st, want_st)
    np.testing.assert_array_equal(got_mr, want_mr)


@pytest.mark.gpu
def test_hip_tabulate_and_rescale_in_one_pass(tmp_path):
    """BASELINE configs[4]: one resident batch, one call — the count tables and the rescaled qualities
    (mdx_tabulate_rescale_device) — against the oracle's tabulation and the oracle's rescaling."""
    import torch

    from mapdamage_amd.engine import DamageEngine
    from oracle import oracle
    from tests.util import assert_tables_equal, oracle_tableset
    _, _, model, corr_prob, _, _ = load(tmp_path)
    ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500,
                            lower_run=3000)
    b = synth.make_reads(ref, 80_000, 23, len_range=(30, 150), paired=True, frac_softclip=0.15, frac_ins=0.06,
                         frac_del=0.06, frac_skip=0.01, with_qual=True, frac_filtered=0.03)
    rng = np.random.default_rng(4)
    b.mtid = b.tid.copy()
    b.mpos = (b.pos + rng.integers(-300, 300, size=b.n)).astype(np.int32)
    libs = [("s", "l")]
    want_tables = oracle_tableset(ref, b, libs, 70, 10, 0)
    want_q, want_mr, want_st = oracle.rescale(ref, b, corr_table(corr_prob, model), model.len5p, model.len3p)
    dev = torch.device("cuda", 0)
    with DamageEngine(libs, 70, 10, 0) as eng:
        eng.set_reference(ref)
        eng.set_rescale_model(model)
        db = eng.upload(b)
        mtid, mpos = torch.from_numpy(b.mtid).to(dev), torch.from_numpy(b.mpos).to(dev)
        qout = torch.zeros(b.seq.shape[0] + 64, dtype=torch.uint8, device=dev)
        mr = torch.zeros(b.n, dtype=torch.float64, device=dev)
        st = torch.zeros(b.n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        eng.rescale_device(db, mtid.data_ptr(), mpos.data_ptr(), qout.data_ptr(), mr.data_ptr(), st.data_ptr(),
                           with_tables=True)
        got_tables = eng.finish()
        db.free()
    assert_tables_equal(got_tables, want_tables)
    np.testing.assert_array_equal(qout.cpu().numpy()[:b.seq.shape[0]], want_q)
    np.testing.assert_array_equal(st.cpu().numpy(), want_st)
    got_mr = mr.cpu().numpy()
    assert np.array_equal(np.isnan(got_mr), np.isnan(want_mr))
    np.testing.assert_array_equal(got_mr[~np.isnan(got_mr)], want_mr[~np.isnan(want_mr)])


@pytest.mark.gpu
@pytest.mark.parametrize("with_tables", [True, False])
def test_hip_rescale_of_a_batch_whose_seq_column_carries_the_min_basequal_mask(tmp_path, with_tables):
    """A context with --min-basequal uploads a 4-bit batch as MDX_SEQ_4BITQ (the mask folded into the nibbles, include/mdx.h).
    The rescaling reads the bases of such a column as the bases they are — rescale.py knows no --min-basequal — and the
    tabulation of the same call counts with the mask (align.py:53-73)."""
    import torch

    from mapdamage_amd.engine import DamageEngine
    from oracle import oracle
    from tests.util import assert_tables_equal, oracle_tableset
    _, _, model, corr_prob, _, _ = load(tmp_path)
    ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500, lower_run=3000)
    b = synth.make_reads(ref, 60_000, 27, len_range=(30, 150), paired=True, frac_softclip=0.15, frac_ins=0.06, frac_del=0.06,
                         frac_skip=0.01, with_qual=True, frac_filtered=0.03)
    rng = np.random.default_rng(5)
    b.mtid = b.tid.copy()
    b.mpos = (b.pos + rng.integers(-300, 300, size=b.n)).astype(np.int32)
    libs = [("s", "l")]
    Q = 20
    want_q, want_mr, want_st = oracle.rescale(ref, b, corr_table(corr_prob, model), model.len5p, model.len3p)
    dev = torch.device("cuda", 0)
    with DamageEngine(libs, 70, 10, Q) as eng:
        eng.set_reference(ref)
        eng.set_rescale_model(model)
        db = eng.upload(b, packed=True)
        from mapdamage_amd.engine import SEQ_4BITQ
        assert db.dev.seq_format == SEQ_4BITQ
        mtid, mpos = torch.from_numpy(b.mtid).to(dev), torch.from_numpy(b.mpos).to(dev)
        qout = torch.zeros(b.seq.shape[0] + 64, dtype=torch.uint8, device=dev)
        mr = torch.zeros(b.n, dtype=torch.float64, device=dev)
        st = torch.zeros(b.n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        eng.rescale_device(db, mtid.data_ptr(), mpos.data_ptr(), qout.data_ptr(), mr.data_ptr(), st.data_ptr(), with_tables=with_tables)
        if not with_tables:
            eng.tabulate(db)
        got_tables = eng.finish()
        db.free()
    assert_tables_equal(got_tables, oracle_tableset(ref, b, libs, 70, 10, Q))
    np.testing.assert_array_equal(qout.cpu().numpy()[:b.seq.shape[0]], want_q)
    np.testing.assert_array_equal(st.cpu().numpy(), want_st)
    got_mr = mr.cpu().numpy()
    assert np.array_equal(np.isnan(got_mr), np.isnan(want_mr))
    np.testing.assert_array_equal(got_mr[~np.isnan(got_mr)], want_mr[~np.isnan(want_mr)])


def one_pass(eng, b, packed=False, patches=False, with_tables=True):
    """mdx_tabulate_rescale_device on the uploaded batch (``packed``: its SEQ column in the 4-bit form) -> (qualities, MR,
    status) as host arrays; the tables stay in the engine.  ``patches``: through mdx_tabulate_rescale_patches_device — the
    rescaled bytes as a list, which must name every byte once, only bytes that change, and applied to the quality column
    (mdx_rescale_expand_device) give the same column; a list too short for them loses the entries beyond it and nothing else."""
    import torch
    dev = torch.device("cuda", 0)
    db = eng.upload(b, packed=packed)
    mtid, mpos = torch.from_numpy(b.mtid).to(dev), torch.from_numpy(b.mpos).to(dev)
    qout = torch.zeros(b.seq.shape[0] + 64, dtype=torch.uint8, device=dev)
    mr = torch.zeros(b.n, dtype=torch.float64, device=dev)
    st = torch.zeros(b.n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    if not patches:
        eng.rescale_device(db, mtid.data_ptr(), mpos.data_ptr(), qout.data_ptr(), mr.data_ptr(), st.data_ptr(), with_tables=with_tables)
        eng.sync()
    else:
        parts, cap = 16, b.n * 26 // 16
        hole = np.uint64(0xFFFFFFFFFFFFFFFF)
        plist = torch.full((parts * cap,), -1, dtype=torch.int64, device=dev)
        count = torch.full((parts,), 12345, dtype=torch.int64, device=dev)
        eng.rescale_patches(db, mtid.data_ptr(), mpos.data_ptr(), plist.data_ptr(), cap, parts, count.data_ptr(), mr.data_ptr(), st.data_ptr(),
                            with_tables=with_tables)
        eng.rescale_expand(db, plist.data_ptr(), cap, parts, count.data_ptr(), qout.data_ptr())
        eng.sync()
        counts = count.cpu().numpy()
        n = int(counts.sum())
        assert n > 0 and int(counts.max()) <= cap
        entries = plist.cpu().numpy().view(np.uint64).reshape(parts, cap)
        for p_ in range(parts):
            assert (entries[p_, int(counts[p_]):] == hole).all()        # nothing behind the entries a part counts
        listed = np.concatenate([entries[p_, :int(counts[p_])] for p_ in range(parts)])
        idx, newq = (listed & np.uint64(0xFFFFFFFF)).astype(np.int64), (listed >> np.uint64(32)).astype(np.int64)
        assert len(np.unique(idx)) == n and idx.max() < b.seq.shape[0]      # every byte once
        assert (newq != b.qual[idx]).all() and newq.max() <= 93             # ... and only bytes that change
        if with_tables:
            # the same launch with parts of half the room of the fullest (the engine's tables would hold the batch twice: reset)
            eng.reset()
            small = int(counts.max()) // 2
            plist.fill_(-1)
            eng.rescale_patches(db, mtid.data_ptr(), mpos.data_ptr(), plist.data_ptr(), small, parts, count.data_ptr(), mr.data_ptr(),
                                st.data_ptr(), with_tables=True)
            eng.sync()
            # (which block takes which tile is settled while the launch runs: the parts fill differently, their sum is the same)
            counts2 = count.cpu().numpy()
            assert int(counts2.sum()) == n and int(counts2.max()) > small
            again = plist.cpu().numpy().view(np.uint64)
            assert (again[parts * small:] == hole).all()
            again = again[:parts * small].reshape(parts, small)
            for p_ in range(parts):
                k_ = min(small, int(counts2[p_]))
                assert (again[p_, :k_] != hole).all() and (again[p_, k_:] == hole).all()
    out = qout.cpu().numpy()[:b.seq.shape[0]], mr.cpu().numpy(), st.cpu().numpy()
    db.free()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("out_form", ["column", "patches"])
@pytest.mark.parametrize("seq_form", ["ascii", "4bit"])
@pytest.mark.parametrize("length,l5,l3,lens", [(70, 12, 12, (25, 160)), (25, 12, 12, (20, 120)), (70, 20, 3, (15, 90)),
                                               (12, 0, 30, (15, 60))])
def test_hip_fused_pass_against_the_oracle_and_the_two_kernel_path(tmp_path, monkeypatch, length, l5, l3, lens, seq_form, out_form):
    """The fused launch (the tabulation kernel rescales the records of its own tile loop and lists the others for the
    rescale kernels): tables, qualities, MR, routing and every summary word — against the oracle, and the summary
    word for word against the two-kernel path on the same batch.  Record lengths on both sides of --length and of
    2 x --length (longer records are listed), records the tabulation drops but the rescaling takes, models with
    uneven windows.  Both forms of the SEQ column: an ASCII one runs the fused ASCII kernel, a 4-bit one the packed fused
    kernel (the records it lists are unpacked for the rescale kernels behind it).  Both forms of the result: a second
    quality column, and the list of the bytes that change (``one_pass``)."""
    packed = seq_form == "4bit"
    patches = out_form == "patches"
    from mapdamage_amd.engine import DamageEngine
    from mapdamage_amd.rescale import RescaleModel
    from oracle import oracle
    from tests.util import assert_tables_equal, oracle_tableset
    rng = np.random.default_rng(100 + length)
    corr_prob = {}
    for p in list(range(1, l5 + 1)) + list(range(-l3, 0)):
        corr_prob[("C", "T", p)] = float(rng.random() * 0.7)
        corr_prob[("G", "A", p)] = float(rng.random() * 0.7)
    model = RescaleModel(corr_prob, l5, l3)
    ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500,
                            lower_run=3000)
    b = synth.make_reads(ref, 70_000, 40 + length, len_range=lens, paired=True, frac_softclip=0.2, frac_ins=0.05,
                         frac_del=0.05, frac_skip=0.01, with_qual=True, frac_filtered=0.03)
    b.mtid = np.where(rng.random(b.n) < 0.9, b.tid, (b.tid + 1) % 2).astype(np.int32)
    b.mpos = (b.pos + rng.integers(-300, 300, size=b.n)).astype(np.int32)
    b.flag = np.where(rng.random(b.n) < 0.4, b.flag & 0xF14, b.flag).astype(np.uint16)      # unpaired: rescaled from both ends
    b.flag = np.where(rng.random(b.n) < 0.05, b.flag | 0x400, b.flag).astype(np.uint16)     # duplicates: rescaled, not counted
    for i in np.flatnonzero(rng.random(b.n) < 0.01):                                        # records without qualities
        b.qual[b.seq_off[i]:b.seq_off[i + 1]] = 0xFF
    libs = [("s", "l")]
    want_tables = oracle_tableset(ref, b, libs, length, 10, 0)
    want_q, want_mr, want_st, want_counts, _ = oracle.rescale_with_subs(ref, b, corr_table(corr_prob, model), l5, l3)
    got = {}
    for fuse in (True, False):
        monkeypatch.setenv("MDX_NO_FUSE", "0" if fuse else "1")
        with DamageEngine(libs, length, 10, 0) as eng:
            eng.set_reference(ref)
            eng.set_rescale_model(model)
            q, mr, st = one_pass(eng, b, packed, patches)
            # (patches: the pass ran twice — the second time into a list too short — behind a reset of the tables)
            assert eng.fused_launches() == ((2 if patches else 1) if fuse else 0)
            assert eng.packed_launches() == ((2 if patches else 1) if packed else 0)
            words = eng.rescale_summary()
            if patches:
                assert (words % 2 == 0).all()
                words = words // 2
            tables = eng.finish()
        assert_tables_equal(tables, want_tables)
        np.testing.assert_array_equal(q, want_q)
        np.testing.assert_array_equal(st, want_st)
        assert np.array_equal(np.isnan(mr), np.isnan(want_mr))
        np.testing.assert_array_equal(mr[~np.isnan(mr)], want_mr[~np.isnan(want_mr)])
        np.testing.assert_array_equal(words[:756], summary_ints_from_oracle(want_counts))
        got[fuse] = words
    np.testing.assert_array_equal(got[True], got[False])


@pytest.mark.gpu
def test_hip_fused_pass_with_two_libraries(tmp_path, monkeypatch):
    """Two libraries in one fused launch (--length 12, fragment lengths up to 128: both images fit next to the fused
    kernel's own tables): a second TC table per library."""
    from mapdamage_amd.engine import DamageEngine
    from oracle import oracle
    from tests.util import assert_tables_equal, oracle_tableset
    _, _, model, corr_prob, _, _ = load(tmp_path)
    ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000)), n_run=500, lower_run=3000)
    b = synth.make_reads(ref, 50_000, 77, len_range=(20, 60), nlib=2, paired=False, frac_softclip=0.2, frac_ins=0.04,
                         frac_del=0.04, with_qual=True, frac_filtered=0.02)
    b.mtid = b.tid.copy(); b.mpos = b.pos.copy()
    libs = [("s", "l0"), ("s", "l1")]
    want_tables = oracle_tableset(ref, b, libs, 12, 10, 0, lgd_max=128)
    want_q, want_mr, want_st, want_counts, _ = oracle.rescale_with_subs(ref, b, corr_table(corr_prob, model), model.len5p, model.len3p)
    monkeypatch.setenv("MDX_NO_FUSE", "0")
    with DamageEngine(libs, 12, 10, 0, lgd_max=128) as eng:
        eng.set_reference(ref)
        eng.set_rescale_model(model)
        q, mr, st = one_pass(eng, b)
        assert eng.fused_launches() == 1
        words = eng.rescale_summary()
        tables = eng.finish()
    assert_tables_equal(tables, want_tables)
    np.testing.assert_array_equal(q, want_q)
    np.testing.assert_array_equal(st, want_st)
    assert np.array_equal(np.isnan(mr), np.isnan(want_mr))
    np.testing.assert_array_equal(mr[~np.isnan(mr)], want_mr[~np.isnan(want_mr)])
    np.testing.assert_array_equal(words[:756], summary_ints_from_oracle(want_counts))


@pytest.mark.gpu
def test_hip_rescale_names_the_record_it_cannot_process(tmp_path):
    """A record running past its contig end (the reference's fetch raises there) is reported with its index."""
    from mapdamage_amd.batch import batch_from_records
    from mapdamage_amd.engine import BadReadError, DamageEngine
    _, _, model, _, _, _ = load(tmp_path)
    ref = synth.small_genome()
    recs = hardclip_records(ref, n=40)
    recs[17] = dict(recs[17], pos=ref.lengths[0] - 10, cigar=[(0, 50)], seq="A" * 50, qual=np.full(50, 30, np.uint8))
    b = batch_from_records(recs, with_qual=True)
    b.mtid = b.tid.copy(); b.mpos = b.pos.copy()
    with DamageEngine([("s", "l")]) as eng:
        eng.set_reference(ref)
        eng.set_rescale_model(model)
        with pytest.raises(BadReadError) as err:
            eng.rescale(b)
    assert err.value.read_index == 17


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [[], ["--host-decode"], ["--host-deflate"]], ids=["device", "host-decode", "host-deflate"])
def test_cli_rescale_only_rewrites_bam(tmp_path, flags):
    """`--rescale-only`: every record written back, new qualities + MR:f on the rescaled ones,
    untouched fields preserved.  Three routes: the records never on the host (inflated, rescaled, written back and deflated in
    HBM: the default), the host decoder with the BGZF writer on the device, the host decoder with zlib."""
    import struct

    from mapdamage_amd import fasta, sam
    from mapdamage_amd.main import main
    ref, batch, model, corr_prob, want_qual, want_mr = load(tmp_path)
    sam.write_bam(tmp_path / "in.bam", batch, ref.names, ref.lengths, [], None)
    fasta.write_fasta(tmp_path / "ref.fa", ref)
    folder = tmp_path / "res"
    folder.mkdir()
    (folder / "Stats_out_MCMC_correct_prob.csv").write_bytes((tmp_path / "Stats_out_MCMC_correct_prob.csv").read_bytes())
    rc = main(["-i", str(tmp_path / "in.bam"), "-r", str(tmp_path / "ref.fa"), "-d", str(folder), "--rescale-only",
               "--rescale-length-5p", "12", "--rescale-length-3p", "10"] + flags)
    assert rc == 0
    assert "gave up" not in (folder / "Runtime_log.txt").read_text()
    out = sam.read_bam(folder / "in.rescaled.bam", keep_raw=True)
    assert out.batch.n == batch.n
    np.testing.assert_array_equal(out.batch.qual, want_qual)
    for k in ("flag", "tid", "pos", "tlen", "cigar", "seq", "mtid", "mpos"):
        np.testing.assert_array_equal(getattr(out.batch, k), getattr(batch, k), err_msg=k)
    assert out.qname == ["r%d" % i for i in range(batch.n)]
    for i, body in enumerate(out.raw):
        if np.isnan(want_mr[i]):
            assert not out.has_mr[i]
        else:
            assert out.has_mr[i] and body[-7:-4] == b"MRf"
            assert struct.unpack("<f", body[-4:])[0] == np.float32(want_mr[i])


@pytest.mark.gpu
def test_cli_rescale_only_tolerates_a_fasta_without_an_unused_contig(tmp_path):
    """The reference's --rescale-only branch (main.py:121-124) makes no .fai / dictionary check: a FASTA that lacks a
    header sequence no record maps to (and has no .fai) still rescales the file."""
    from mapdamage_amd import fasta, sam
    from mapdamage_amd.batch import Reference
    from mapdamage_amd.main import main
    ref, batch, model, corr_prob, want_qual, want_mr = load(tmp_path)
    used = sorted(set(int(t) for t in batch.tid if t >= 0))
    spare = [t for t in range(len(ref.names)) if t not in used]
    names = list(ref.names) + ["unused_extra"]
    lengths = list(ref.lengths) + [1234]
    sam.write_bam(tmp_path / "in.bam", batch, names, lengths, [], None)
    keep = [t for t in range(len(ref.names)) if t not in spare[:1]]
    fasta.write_fasta(tmp_path / "ref.fa", Reference([ref.names[t] for t in keep], [ref.seqs[t] for t in keep]))
    fai = tmp_path / "ref.fa.fai"
    if fai.exists():
        fai.unlink()
    folder = tmp_path / "res"
    folder.mkdir()
    (folder / "Stats_out_MCMC_correct_prob.csv").write_bytes((tmp_path / "Stats_out_MCMC_correct_prob.csv").read_bytes())
    rc = main(["-i", str(tmp_path / "in.bam"), "-r", str(tmp_path / "ref.fa"), "-d", str(folder), "--rescale-only",
               "--rescale-length-5p", "12", "--rescale-length-3p", "10"])
    assert rc == 0
    out = sam.read_bam(folder / "in.rescaled.bam", keep_raw=True)
    np.testing.assert_array_equal(out.batch.qual, want_qual)


@pytest.mark.gpu
def test_rescale_on_device_refuses_a_record_that_has_an_mr_tag(tmp_path):
    """rescale.py:277-278: a record that is to be rescaled and carries an MR tag already stops the pass, by name — on the path
    that never brings the records to the host too (the write-back kernel looks at the tags of the records it extends)."""
    from mapdamage_amd import sam
    from mapdamage_amd.engine import DamageEngine
    from mapdamage_amd.rescale import rescale_bam, rescale_bam_on_device
    ref, batch, model, corr_prob, want_qual, want_mr = load(tmp_path)
    sam.write_bam(tmp_path / "in.bam", batch, ref.names, ref.lengths, [], None)
    with DamageEngine([("*", "*")]) as eng:
        rescale_bam_on_device(eng, ref, tmp_path / "in.bam", tmp_path / "once.bam", model)
    first = next(i for i in range(batch.n) if not np.isnan(want_mr[i]))
    for fn in (rescale_bam_on_device, rescale_bam):
        with DamageEngine([("*", "*")]) as eng:
            with pytest.raises(SystemExit, match="Read: r%d already has a MR tag" % first):
                fn(eng, ref, tmp_path / "once.bam", tmp_path / "twice.bam", model)
