"""GPU parity: the HIP path (through the C-ABI) against the golden vectors produced by the
reference and against the C oracle on seeded inputs.  Integer tables: bit-exact."""

import pathlib

import numpy as np
import pytest

from mapdamage_amd import synth
from mapdamage_amd.batch import batch_from_records
from tests.util import Golden, assert_tables_equal, golden_names, oracle_tableset

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["ascii", "4bit"])
def seq_form(request):
    """Every case of this module runs through both forms of the SEQ column (include/mdx.h MDX_SEQ_*): as ASCII, and
    packed to 4 bits on the host — the packed kernels where the launch has the fast geometry (with --min-basequal the
    masked form), an ASCII scratch copy made by the library where it has not (the generic path)."""
    from mapdamage_amd.engine import DamageEngine
    old = DamageEngine.default_packed
    DamageEngine.default_packed = request.param == "4bit"
    yield request.param
    DamageEngine.default_packed = old


def run_engine(ref, batch, libraries, length, around, minqual=0, lgd_max=65536, resident=False,
               splits=1):
    from mapdamage_amd.engine import DamageEngine
    with DamageEngine(libraries, length, around, minqual, lgd_max=lgd_max) as eng:
        eng.set_reference(ref)
        for k in range(splits):
            part = batch.shard(k, splits) if splits > 1 else batch
            if resident:
                dev = eng.upload(part)
                eng.tabulate(dev)
                eng.sync()
                dev.free()
            else:
                eng.tabulate(part)
        # the packed kernel is what runs for a 4-bit column in a tabulation with the fast geometry (ONE launch per call whatever
        # the number of libraries — a library per pool of blocks over the records bucketed by library; with --min-basequal its masked form),
        # and only then
        plain = eng.table_mode == "lds" and length + around <= 248
        want = splits if (DamageEngine.default_packed and plain and batch.n) else 0
        assert eng.packed_launches() == want
        # (several libraries: a resident batch brings the bucketed columns, a host batch is sorted inside its launch)
        assert eng.libsorts() == (want if len(libraries) > 1 and not resident else 0)
        return eng.finish()


@pytest.mark.parametrize("name", golden_names())
def test_hip_matches_reference_golden(name):
    g = Golden(name)
    ts = run_engine(g.ref, g.batch, g.libraries, g.length, g.around, g.minqual, lgd_max=4096)
    g.check(ts)


@pytest.mark.parametrize("name", ["config1_L70_A10_Q20", "edge_L8_A3_Q25"])
def test_hip_resident_and_split_batches(name):
    g = Golden(name)
    ts = run_engine(g.ref, g.batch, g.libraries, g.length, g.around, g.minqual, lgd_max=4096,
                    resident=True, splits=3)
    g.check(ts)


CASES = [
    dict(n=200_000, seed=2, kw=dict(read_len=100), L=70, A=10, Q=0),
    dict(n=100_000, seed=3, kw=dict(read_len=100, paired=True, frac_softclip=0.10, frac_ins=0.04,
                                    frac_del=0.04, frac_skip=0.002, frac_hardclip=0.001), L=70, A=10, Q=0),
    dict(n=100_000, seed=4, kw=dict(len_range=(35, 150), paired=True, frac_softclip=0.10,
                                    frac_ins=0.04, frac_del=0.04, frac_skip=0.002,
                                    frac_hardclip=0.001, nlib=3, frac_filtered=0.05,
                                    frac_n_base=0.02), L=70, A=10, Q=0),
    dict(n=60_000, seed=5, kw=dict(len_range=(20, 90), frac_softclip=0.2, frac_ins=0.1, frac_del=0.1,
                                   frac_skip=0.02, with_qual=True, nlib=2), L=25, A=4, Q=20),
    dict(n=30_000, seed=6, kw=dict(len_range=(30, 300), frac_softclip=0.2, frac_ins=0.1, frac_del=0.1,
                                   with_qual=True), L=150, A=30, Q=12),
    dict(n=20_000, seed=7, kw=dict(read_len=50, nlib=1), L=1, A=0, Q=0),
    # two records per wavefront step (A + L = 112: 28 lanes per record)
    dict(n=60_000, seed=9, kw=dict(len_range=(60, 200), frac_softclip=0.1, frac_ins=0.03, frac_del=0.03,
                                   with_qual=True), L=100, A=12, Q=10),
    # A + L > 248: no 8-byte-lane path, every record walks its CIGAR (tables still in the LDS)
    dict(n=5_000, seed=10, kw=dict(len_range=(50, 400), frac_softclip=0.1, frac_ins=0.03, frac_del=0.03),
         L=260, A=5, Q=0),
]


@pytest.fixture(scope="module")
def mid_genome():
    return synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)),
                             n_run=500, lower_run=3000)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d_L%d_Q%d" % (c["seed"], c["L"], c["Q"]))
def test_hip_matches_oracle_seeded(case, mid_genome):
    batch = synth.make_reads(mid_genome, case["n"], case["seed"], **case["kw"])
    nlib = case["kw"].get("nlib", 1)
    libs = [("S%d" % i, "L%d" % i) for i in range(nlib)]
    want = oracle_tableset(mid_genome, batch, libs, case["L"], case["A"], case["Q"])
    got = run_engine(mid_genome, batch, libs, case["L"], case["A"], case["Q"])
    assert_tables_equal(got, want)
    assert got.misincorporation_text() == want.misincorporation_text()


def test_hip_sorted_batch_with_gap_run(mid_genome):
    """Coordinate-sorted batch: the reads over the genome's N-run (every byte an event) are consecutive
    records; tiles are dealt round-robin to the wavefronts."""
    batch = synth.make_reads(mid_genome, 150_000, 13, read_len=100)
    batch = synth._permute_fixed(batch, np.lexsort((batch.pos, batch.tid)))
    libs = [("s", "l")]
    want = oracle_tableset(mid_genome, batch, libs, 70, 10, 0)
    got = run_engine(mid_genome, batch, libs, 70, 10, 0, resident=True)
    assert_tables_equal(got, want)


@pytest.mark.parametrize("kw", [dict(frac_ins=0.5, frac_del=0.5), dict(frac_softclip=1.0, frac_ins=0.3, frac_del=0.3, frac_skip=0.2),
                                dict(len_range=(20, 69), frac_ins=0.2, frac_del=0.2)])
def test_hip_lists_full_in_every_round(kw, mid_genome):
    """Batches none of whose records is a plain one: every tile appends 63 entries to the wavefront's rings (the lists of
    single insertions, single deletions, partial records; the ring of records handed to the general pass), which wrap
    within a launch and are emptied round by round (csrc/mdx_internal.h: MDX_LIST_RING, MDX_ROUND_TILES, MDX_DRING)."""
    batch = synth.make_reads(mid_genome, 400_000, 31, read_len=100, **kw)
    libs = [("s", "l")]
    want = oracle_tableset(mid_genome, batch, libs, 70, 10, 0)
    got = run_engine(mid_genome, batch, libs, 70, 10, 0, resident=True)
    assert_tables_equal(got, want)


def test_hip_every_column_an_event(mid_genome):
    """Reads whose every base differs from the reference (A <-> C, G <-> T): every lane of every step queues an event, the
    packed kernel's queue overflows in every group of steps — in the first and in the second of a pair of groups alike
    (csrc/mdx_kernels.hip: run(), `ovf`) — and the run drains and starts again from the step that did not fit."""
    batch = synth.make_reads(mid_genome, 60_000, 29, read_len=100, frac_softclip=0.1, frac_ins=0.03, frac_del=0.03)
    swap = np.arange(256, dtype=np.uint8)
    for a, b in (("A", "C"), ("G", "T")):
        swap[ord(a)], swap[ord(b)] = ord(b), ord(a)
    batch.seq[:] = swap[batch.seq]
    libs = [("s", "l")]
    want = oracle_tableset(mid_genome, batch, libs, 70, 10, 0)
    got = run_engine(mid_genome, batch, libs, 70, 10, 0, resident=True)
    assert_tables_equal(got, want)


@pytest.mark.parametrize("Q", [15, 0])
@pytest.mark.parametrize("phase", [1, 2, 3])
def test_hip_unaligned_column_pointers(phase, Q, mid_genome, seq_form):
    """SEQ / QUAL device pointers that are not dword-aligned (the kernel aligns its window loads down and
    folds the pointer phase into the lane offsets)."""
    import torch
    from mapdamage_amd.engine import DamageEngine
    batch = synth.make_reads(mid_genome, 30_000, 31, len_range=(30, 140), with_qual=True, frac_softclip=0.1,
                             frac_ins=0.03, frac_del=0.03)
    libs = [("s", "l")]
    want = oracle_tableset(mid_genome, batch, libs, 70, 10, Q)
    with DamageEngine(libs, 70, 10, Q) as eng:
        eng.set_reference(mid_genome)
        dev = eng.upload(batch)
        n = int(batch.seq.shape[0])
        tseq = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        tqual = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        if seq_form == "4bit":
            from mapdamage_amd.engine import pack_seq
            pk = pack_seq(batch.seq)
            tseq[phase:phase + pk.shape[0]] = torch.from_numpy(pk).cuda()
        else:
            tseq[phase:phase + n] = torch.from_numpy(batch.seq).cuda()
        tqual[phase:phase + n] = torch.from_numpy(batch.qual).cuda()
        torch.cuda.synchronize()
        own = (dev.dev.seq, dev.dev.qual, dev.dev.seq_format)
        dev.dev.seq, dev.dev.qual = tseq.data_ptr() + phase, tqual.data_ptr() + phase
        if seq_form == "4bit":
            dev.dev.seq_format = 1      # (the caller's own MDX_SEQ_4BIT column: with -Q the mask is folded in front of the launch)
        try:
            eng.tabulate(dev)
            got = eng.finish()
        finally:
            dev.dev.seq, dev.dev.qual, dev.dev.seq_format = own
            dev.free()
    assert_tables_equal(got, want)


@pytest.mark.parametrize("nlib,Q", [(2, 0), (3, 0), (7, 15), (8, 0)])
def test_hip_library_groups_match_oracle(nlib, Q, mid_genome):
    """More libraries than fit the LDS at once: one launch per group of libraries, every launch counting the
    records of its group (tables stay in the LDS)."""
    from mapdamage_amd.engine import DamageEngine
    batch = synth.make_reads(mid_genome, 40_000, 8, len_range=(30, 120), nlib=nlib, frac_softclip=0.1,
                             frac_ins=0.05, frac_del=0.05, with_qual=True, paired=True)
    libs = [("S%d" % i, "L%d" % i) for i in range(nlib)]
    want = oracle_tableset(mid_genome, batch, libs, 70, 10, Q, lgd_max=300)
    with DamageEngine(libs, 70, 10, Q, lgd_max=300) as eng:     # (lengths >= 300 take the overflow list)
        assert eng.table_mode == "lds"
        eng.set_reference(mid_genome)
        eng.tabulate(batch)
        eng.tabulate(batch.slice(0, 1000))
        got = eng.finish()
        # (a 4-bit column: one launch per call — a library per pool of blocks over the records bucketed by library — not one per library)
        assert eng.packed_launches() == (2 if DamageEngine.default_packed else 0)
    want2 = oracle_tableset(mid_genome, batch.slice(0, 1000), libs, 70, 10, Q, lgd_max=300)
    np.testing.assert_array_equal(got.mis, want.mis + want2.mis)
    np.testing.assert_array_equal(got.comp, want.comp + want2.comp)
    assert got.n_kept == want.n_kept + want2.n_kept
    merged = {}
    for key in want.lgd_sparse() + want2.lgd_sparse():
        merged[key[:-1]] = merged.get(key[:-1], 0) + key[-1]
    assert sorted(got.lgd_sparse()) == sorted(k + (v,) for k, v in merged.items())


@pytest.mark.parametrize("nlib", [2, 5, 40, 300])
def test_hip_one_pass_over_the_libraries_of_a_resident_batch(nlib, mid_genome):
    """reader.py:47-50, statistics.py:12-20: the tables are keyed by library and a file interleaves the libraries.  A resident
    4-bit batch brings its columns bucketed by library (mdx_batch::libsort) and every call is one launch of the packed kernel
    per sixty-four libraries — a library per pool of blocks, over its own records —, libraries without a single record, filtered
    records and skewed library sizes included; accumulating calls, and the same batch through the in-launch sort."""
    from mapdamage_amd.engine import DamageEngine
    batch = synth.make_reads(mid_genome, 30_000, 8 + nlib, len_range=(30, 120), nlib=nlib, frac_softclip=0.1, frac_ins=0.05,
                             frac_del=0.05, paired=True, frac_filtered=0.05)
    rng = np.random.default_rng(nlib)
    # skewed: half of the records in library 1, nothing in the last library
    batch.lib[rng.random(batch.n) < 0.5] = 1
    batch.lib[batch.lib == nlib - 1] = 0
    libs = [("S%d" % i, "L%d" % i) for i in range(nlib)]
    want = oracle_tableset(mid_genome, batch, libs, 70, 10, 0, lgd_max=300)
    with DamageEngine(libs, 70, 10, 0, lgd_max=300) as eng:
        eng.set_reference(mid_genome)
        dev = eng.upload(batch, packed=True)
        assert dev.dev.libsort
        eng.tabulate(dev)
        eng.tabulate(dev)
        eng.sync()
        launches = eng.packed_launches()
        assert eng.libsorts() == 0
        # (a batch that does not bring the bucketed columns: sorted inside the launch)
        view = type(dev.dev)()
        import ctypes
        ctypes.memmove(ctypes.byref(view), ctypes.byref(dev.dev), ctypes.sizeof(view))
        view.libsort = None
        eng.tabulate_view(view)
        eng.sync()
        assert eng.libsorts() == 1
        got = eng.finish()
        dev.free()
    assert launches % 2 == 0 and launches // 2 <= (nlib + 15) // 16 + 1      # not one launch per library
    np.testing.assert_array_equal(got.mis, 3 * want.mis)
    np.testing.assert_array_equal(got.comp, 3 * want.comp)
    assert got.n_kept == 3 * want.n_kept
    assert sorted(got.lgd_sparse()) == sorted(k[:-1] + (3 * k[-1],) for k in want.lgd_sparse())


def test_hip_one_pass_over_the_libraries_with_min_basequal(mid_genome):
    from mapdamage_amd.engine import DamageEngine
    batch = synth.make_reads(mid_genome, 40_000, 77, len_range=(30, 120), nlib=6, frac_softclip=0.1, frac_ins=0.05,
                             frac_del=0.05, with_qual=True, paired=True, frac_filtered=0.03)
    libs = [("S%d" % i, "L%d" % i) for i in range(6)]
    want = oracle_tableset(mid_genome, batch, libs, 70, 10, 20, lgd_max=300)
    with DamageEngine(libs, 70, 10, 20, lgd_max=300) as eng:
        eng.set_reference(mid_genome)
        dev = eng.upload(batch, packed=True)
        assert dev.dev.libsort and dev.dev.seq_format == 2      # (MDX_SEQ_4BITQ: the mask is in the column, ordered with it)
        eng.tabulate(dev)
        got = eng.finish()
        assert eng.packed_launches() == 1 and eng.libsorts() == 0
        dev.free()
    assert_tables_equal(got, want)


def test_hip_library_id_beyond_the_last_in_a_resident_batch(mid_genome):
    """The sort gives such a record no place; every launch over the bucketed columns reports it, with the caller's base."""
    from mapdamage_amd.engine import BadReadError, DamageEngine
    batch = synth.make_reads(mid_genome, 5_000, 8, read_len=60, nlib=5)
    batch.lib[1234] = 7
    batch.lib[4321] = 9
    with DamageEngine([("S%d" % i, "L") for i in range(5)]) as eng:
        eng.set_reference(mid_genome)
        dev = eng.upload(batch, packed=True)
        eng.tabulate(dev, record_base=100)
        with pytest.raises(BadReadError) as err:
            eng.sync()
        assert err.value.read_index == 1334
        dev.free()


def test_hip_launch_scratch_does_not_grow_with_the_batch():
    """The per-wavefront lists of a launch are rings of a fixed size (81 KB per wavefront: the kernel works in rounds of 14
    tiles and empties its lists at the end of each) — a resident batch of 40 M records, as large as 32-bit SEQ offsets
    allow for 100-base reads, is tabulated with a few hundred megabytes of scratch (round 4: 180 bytes per record, 7 GB),
    and its tables are those of its sixteen identical parts."""
    import torch
    from mapdamage_amd.batch import concat_batches
    from mapdamage_amd.engine import DamageEngine
    from oracle import oracle
    ref = synth.make_genome()
    part = synth.config3_batch(ref, 2_500_000, seed=77)     # (in this process: the forked generator wants an untouched GPU)
    want, _ = oracle.tabulate_parallel(ref, part, 1, 70, 10, 0, lgd_max=4096)
    big = concat_batches([part] * 16)
    assert big.n == 40_000_000 and int(big.seq_off[-1]) == 4_000_000_000
    with DamageEngine([("s", "l")], 70, 10, 0, lgd_max=4096) as eng:
        eng.set_reference(ref)
        dev = eng.upload(big, packed=True)
        del big
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        eng.tabulate(dev)
        eng.sync()
        free1, _ = torch.cuda.mem_get_info()
        got = eng.finish()
        assert eng.packed_launches() == 1
        dev.free()
    assert free0 - free1 < (1 << 30), "launch scratch: %d MiB" % ((free0 - free1) >> 20)
    np.testing.assert_array_equal(got.mis, 16 * want["mis"])
    np.testing.assert_array_equal(got.comp, 16 * want["comp"])
    np.testing.assert_array_equal(got.lgd, 16 * want["lgd"])
    assert got.n_kept == 16 * want["n_kept"]


def test_hip_library_id_beyond_the_last_is_an_error(mid_genome):
    from mapdamage_amd.engine import BadReadError, DamageEngine
    batch = synth.make_reads(mid_genome, 5_000, 8, read_len=60, nlib=5)
    batch.lib[1234] = 7
    with DamageEngine([("S%d" % i, "L") for i in range(5)]) as eng:
        eng.set_reference(mid_genome)
        with pytest.raises(BadReadError) as err:     # a host batch reports at once, with the index within the batch
            eng.tabulate(batch)
        assert err.value.read_index == 1234
        with pytest.raises(BadReadError) as err:     # ... and the context stays in error until reset
            eng.sync()
    assert "1234" in str(err.value)


def test_hip_global_atomic_fallback_matches_oracle(mid_genome):
    """Tables of a single library too large for the LDS (very large --length) take the global-atomic path."""
    from mapdamage_amd.engine import DamageEngine
    batch = synth.make_reads(mid_genome, 20_000, 8, len_range=(30, 900), nlib=2, frac_softclip=0.1,
                             frac_ins=0.05, frac_del=0.05, with_qual=True)
    libs = [("S%d" % i, "L%d" % i) for i in range(2)]
    want = oracle_tableset(mid_genome, batch, libs, 700, 10, 15)
    with DamageEngine(libs, 700, 10, 15) as eng:
        assert eng.table_mode == "global"
        eng.set_reference(mid_genome)
        eng.tabulate(batch)
        got = eng.finish()
    assert_tables_equal(got, want)


@pytest.mark.parametrize("Q", [0, 20])
def test_hip_path_of_a_reference_of_4_gbases_and_more(mid_genome, monkeypatch, Q):
    """A reference of 4 Gbases and more (with its guard bands) takes 64-bit window offsets and the generic CIGAR walk for
    every record (`FAST = false`) — forced here at the default geometry (MDX_FORCE_REF64, a test-only switch: no test
    can hold a 4 Gb genome), against the oracle, with and without --min-basequal, from both forms of the SEQ column."""
    from mapdamage_amd.engine import DamageEngine
    batch = synth.make_reads(mid_genome, 60_000, 81, len_range=(25, 160), nlib=2, paired=True, frac_softclip=0.15,
                             frac_ins=0.06, frac_del=0.06, frac_skip=0.01, frac_hardclip=0.01, with_qual=True, frac_filtered=0.03)
    libs = [("S%d" % i, "L%d" % i) for i in range(2)]
    want = oracle_tableset(mid_genome, batch, libs, 70, 10, Q)
    monkeypatch.setenv("MDX_FORCE_REF64", "1")
    for packed in (False, True):
        with DamageEngine(libs, 70, 10, Q) as eng:
            assert eng.table_mode == "lds"
            eng.set_reference(mid_genome)
            eng.tabulate(batch, packed=packed)
            got = eng.finish()
            assert eng.packed_launches() == 0          # (the packed kernel reads 32-bit offsets: not this path)
        assert_tables_equal(got, want)
    monkeypatch.delenv("MDX_FORCE_REF64")
    with DamageEngine(libs, 70, 10, Q) as eng:
        eng.set_reference(mid_genome)
        eng.tabulate(batch, packed=True)
        got = eng.finish()
        assert eng.packed_launches() == 1          # (two libraries, one launch: half of the pools each)
    assert_tables_equal(got, want)


def test_hip_min_basequal_mask_in_the_column_or_folded_per_launch(mid_genome):
    """The packed masked kernel reads the mask from the SEQ column itself (MDX_SEQ_4BITQ, include/mdx.h: a base whose quality
    is below the threshold is the complement of its code; align.py:65-71).  A batch uploaded to a context with a
    --min-basequal is such a column; a MDX_SEQ_4BIT batch with qualities (a caller's own: here uploaded through a context
    without a threshold) is folded into a scratch column in front of every launch — from its qualities, or from the
    caller's bitmap of them (mdx_batch::lowq).  All against the oracle, with half of the bases masked and with none."""
    import ctypes

    from mapdamage_amd.batch import ReadBatch
    from mapdamage_amd.engine import DamageEngine, MdxBatch
    batch = synth.make_reads(mid_genome, 50_000, 17, len_range=(25, 160), paired=True, frac_softclip=0.15, frac_ins=0.06,
                             frac_del=0.06, frac_skip=0.01, with_qual=True, frac_filtered=0.03)
    libs = [("s", "l")]
    with DamageEngine(libs, 70, 10, 0) as eng0:
        eng0.set_reference(mid_genome)
        plain = eng0.upload(batch, packed=True)
        assert plain.dev.seq_format == 1 and plain.dev.qual
        for Q in (20, 1):
            want = oracle_tableset(mid_genome, batch, libs, 70, 10, Q)
            with DamageEngine(libs, 70, 10, Q) as eng:
                eng.set_reference(mid_genome)
                db = eng.upload(batch, packed=True)
                assert db.dev.seq_format == 2 and not db.dev.lowq       # (folded at upload: the context has a --min-basequal)
                eng.tabulate(db)
                got_resident = eng.finish()
                assert eng.packed_launches() == 1
                eng.reset()
                view = MdxBatch()
                ctypes.memmove(ctypes.byref(view), ctypes.byref(plain.dev), ctypes.sizeof(MdxBatch))
                eng.tabulate_view(view)                 # (folded in front of the launch, from the quality column)
                got_per_launch = eng.finish()
                assert eng.packed_launches() == 2
                eng.reset()
                # ... and from the caller's bitmap (its bytes brought to the device as the `seq` column of an ASCII upload)
                bits = np.packbits(batch.qual < Q, bitorder="little")
                carrier = ReadBatch(np.zeros(1, np.uint16), np.zeros(1, np.uint16), np.zeros(1, np.int32), np.zeros(1, np.int32),
                                    np.zeros(1, np.int32), np.array([0, 0], np.uint32), np.zeros(0, np.uint32),
                                    np.array([0, bits.shape[0]], np.uint32), bits, None)
                dbits = eng.upload(carrier, packed=False)
                view.lowq = dbits.dev.seq
                view.qual = plain.dev.qual
                eng.tabulate_view(view)
                got_bitmap = eng.finish()
                dbits.free()
                db.free()
            assert_tables_equal(got_resident, want)
            assert_tables_equal(got_per_launch, want)
            assert_tables_equal(got_bitmap, want)
        # (a context without a threshold refuses a column that carries one)
        with DamageEngine(libs, 70, 10, 20) as eng:
            eng.set_reference(mid_genome)
            db = eng.upload(batch, packed=True)
            view = MdxBatch()
            ctypes.memmove(ctypes.byref(view), ctypes.byref(db.dev), ctypes.sizeof(MdxBatch))
            from mapdamage_amd.engine import MdxError
            with pytest.raises(MdxError, match="MDX_SEQ_4BITQ"):
                eng0.tabulate_view(view)
            db.free()
            # a caller's own batch folded ONCE (mdx_batch_fold): in place, MDX_SEQ_4BITQ from then on — every launch behind it is
            # the resident batch's launch
            view = MdxBatch()
            ctypes.memmove(ctypes.byref(view), ctypes.byref(plain.dev), ctypes.sizeof(MdxBatch))
            with pytest.raises(MdxError, match="no --min-basequal"):
                eng0.fold(view)
            eng.fold(view)
            assert view.seq_format == 2
            eng.fold(view)                          # (again: nothing to do)
            want = oracle_tableset(mid_genome, batch, libs, 70, 10, 20)
            for _ in range(2):
                eng.reset()
                eng.tabulate_view(view)
                assert_tables_equal(eng.finish(), want)
        plain.free()


def test_hip_rejects_alignment_past_contig_end():
    from mapdamage_amd.engine import BadReadError, DamageEngine
    ref = synth.small_genome()
    n_last = ref.lengths[-1]
    recs = [dict(flag=0, tid=0, pos=10, cigar=[(0, 20)], seq="A" * 20, qual=None, lib=0, tlen=0),
            dict(flag=0, tid=2, pos=n_last - 10, cigar=[(0, 25)], seq="A" * 25, qual=None, lib=0, tlen=0)]
    with DamageEngine([("s", "l")]) as eng:
        eng.set_reference(ref)
        dev = eng.upload(batch_from_records(recs))   # a resident batch is only enqueued: the error surfaces at finish
        eng.tabulate(dev)
        with pytest.raises(BadReadError) as e:
            eng.finish()
        assert e.value.read_index == 1
        dev.free()


def test_hip_empty_and_all_filtered_batches():
    from mapdamage_amd.engine import DamageEngine
    ref = synth.small_genome()
    recs = [dict(flag=0x4, tid=0, pos=10, cigar=[(0, 20)], seq="A" * 20, qual=None, lib=0, tlen=0),
            dict(flag=0x400, tid=0, pos=10, cigar=[(0, 20)], seq="A" * 20, qual=None, lib=0, tlen=0)]
    with DamageEngine([("s", "l")]) as eng:
        eng.set_reference(ref)
        eng.tabulate(batch_from_records([]))
        eng.tabulate(batch_from_records(recs))
        ts = eng.finish()
    assert ts.n_kept == 0 and int(ts.mis.sum()) == 0 and int(ts.comp.sum()) == 0


def test_hip_accumulate_is_linear(mid_genome):
    """Size-independent property: tabulating a batch twice doubles every counter."""
    from mapdamage_amd.engine import DamageEngine
    batch = synth.make_reads(mid_genome, 50_000, 12, read_len=100, frac_softclip=0.1, frac_ins=0.03)
    with DamageEngine([("s", "l")]) as eng:
        eng.set_reference(mid_genome)
        dev = eng.upload(batch)
        eng.tabulate(dev)
        one = eng.finish()
        eng.tabulate(dev)
        two = eng.finish()
        eng.reset()
        eng.tabulate(dev)
        again = eng.finish()
        dev.free()
    np.testing.assert_array_equal(two.mis, 2 * one.mis)
    np.testing.assert_array_equal(two.comp, 2 * one.comp)
    np.testing.assert_array_equal(two.lgd, 2 * one.lgd)
    assert_tables_equal(again, one)


@pytest.mark.parametrize("fmt", ["sam", "bam"])
def test_cli_end_to_end_byte_identical_outputs(tmp_path, fmt):
    """`python -m mapdamage_amd -i x.bam -r ref.fa -Q 20` writes the reference's three tables."""
    from mapdamage_amd import fasta, sam
    from mapdamage_amd.main import main
    g = Golden("config1_L70_A10_Q20")
    rgs = [{"ID": "rg%d" % i, "SM": s, "LB": l} for i, (s, l) in enumerate(g.meta["libraries"])]
    raw_lib = np.load(str(pathlib.Path(__file__).parent / "golden" / "config1_L70_A10_Q20.npz"))["lib"]
    rg_of = ["rg%d" % int(l) for l in raw_lib]
    path = tmp_path / ("in." + fmt)
    (sam.write_sam if fmt == "sam" else sam.write_bam)(path, g.batch, g.ref.names, g.ref.lengths, rgs, rg_of)
    fasta.write_fasta(tmp_path / "ref.fa", g.ref)
    out = tmp_path / "res"
    assert main(["-i", str(path), "-r", str(tmp_path / "ref.fa"), "-d", str(out), "-Q", "20", "--no-stats"]) == 0
    for name in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt"):
        assert (out / name).read_text() == g.txt[name], name
    assert (out / "Runtime_log.txt").exists()
    if fmt == "bam":
        # the same file decoded in ~4 KiB chunks (records and BGZF blocks straddle the borders) and in one piece
        for chunk_mb in ("0.004", "0"):
            out2 = tmp_path / ("res" + chunk_mb)
            assert main(["-i", str(path), "-r", str(tmp_path / "ref.fa"), "-d", str(out2), "-Q", "20", "--no-stats",
                         "--chunk-mb", chunk_mb]) == 0
            for name in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt"):
                assert (out2 / name).read_text() == g.txt[name], (chunk_mb, name)


@pytest.mark.parametrize("config", [2, 3, 4])
def test_hip_matches_oracle_at_scale(config):
    """Survey configs 2/3/4 on the 10 Mb benchmark genome, 1 M records each, through a resident
    batch tabulated twice: the doubled tables must equal twice the oracle's (parity + linearity)."""
    from mapdamage_amd.engine import DamageEngine
    ref = synth.make_genome()
    make = {2: synth.config2_batch, 3: synth.config3_batch, 4: synth.config4_batch}[config]
    batch = make(ref, 1_000_000)
    libs = [("synthetic", "lib1")]
    want = oracle_tableset(ref, batch, libs, 70, 10, 0, lgd_max=4096)
    with DamageEngine(libs, 70, 10, 0, lgd_max=4096) as eng:
        eng.set_reference(ref)
        dev = eng.upload(batch)
        eng.tabulate(dev)
        eng.tabulate(dev)
        got = eng.finish()
        dev.free()
    np.testing.assert_array_equal(got.mis, 2 * want.mis)
    np.testing.assert_array_equal(got.comp, 2 * want.comp)
    np.testing.assert_array_equal(got.lgd, 2 * want.lgd)
    assert got.n_kept == 2 * want.n_kept


def _indel_records(ref, n, seed, max_gap=130, with_qual=False):
    """Records [H][S] aM g{I|D} bM [S][H] (and some with two indels / an N) with indels of 1..max_gap bases anywhere
    in the read, 5 % substitutions and a few N bases: the shapes the single-indel fast path splits into a near and
    a far entry, its limits (|n0 - nq| <= 127, A + |n0 - nq| <= 248) and its fallbacks (the CIGAR walk)."""
    rng = np.random.default_rng(seed)
    bases, offs = ref.concat()
    upper = bases & np.uint8(0xDF)
    lens = list(ref.lengths)
    gaps = [1, 2, 3, 5, 8, 17, 56, 57, 100, 126, 127, 128, max_gap]
    recs = []
    for i in range(n):
        tid = int(rng.integers(0, 2))
        a, b = int(rng.integers(1, 130)), int(rng.integers(1, 130))
        g = int(gaps[rng.integers(0, len(gaps))]) if rng.random() < 0.5 else int(rng.integers(1, 4))
        kind = int(rng.integers(0, 10))          # 0-3 I, 4-7 D, 8 two indels, 9 indel + N
        ops = [(0, a), (1 if kind < 4 or kind == 8 else 2, g), (0, b)]
        if kind == 8:
            ops += [(2, int(rng.integers(1, 4))), (0, int(rng.integers(1, 40)))]
        if kind == 9:
            ops += [(3, int(rng.integers(20, 300))), (0, int(rng.integers(1, 40)))]
        span = sum(ln for op, ln in ops if op in (0, 2, 3))
        margin = 0 if rng.random() < 0.03 else 260   # a few at the very start of a contig (incomplete flank)
        pos = int(rng.integers(margin, lens[tid] - span - 260))
        seq, r = [], offs[tid] + pos
        for op, ln in ops:
            if op == 0:
                seq.append(upper[r:r + ln].copy()); r += ln
            elif op == 1:
                seq.append(rng.choice(np.frombuffer(b"ACGT", np.uint8), ln))
            else:
                r += ln
        sl, sr = (int(rng.integers(1, 9)) if rng.random() < 0.2 else 0 for _ in range(2))
        seq = np.concatenate([rng.choice(np.frombuffer(b"ACGT", np.uint8), sl)] + seq +
                             [rng.choice(np.frombuffer(b"ACGT", np.uint8), sr)])
        mut = rng.random(seq.shape[0])
        seq = np.where(mut < 0.05, rng.choice(np.frombuffer(b"ACGTN", np.uint8), seq.shape[0]), seq)
        cig = ([(5, 3)] if rng.random() < 0.05 else []) + ([(4, sl)] if sl else []) + ops + \
              ([(4, sr)] if sr else []) + ([(5, 2)] if rng.random() < 0.05 else [])
        qual = None
        if with_qual:   # a fifth of the bases below Phred 20, a few records without qualities
            qual = np.where(rng.random(seq.shape[0]) < 0.2, rng.integers(2, 20, seq.shape[0]),
                            rng.integers(20, 42, seq.shape[0])).astype(np.uint8)
            if rng.random() < 0.03:
                qual = None
        recs.append(dict(flag=int(rng.choice([0, 16])), tid=tid, pos=pos, cigar=cig, seq=seq.tobytes().decode(),
                         qual=qual, lib=int(rng.integers(0, 2)), tlen=0))
    return recs


@pytest.mark.parametrize("L,A", [(70, 10), (8, 3), (1, 0), (150, 30), (100, 12), (240, 8), (30, 200), (120, 128),
                                 (250, 10)])
def test_hip_single_indel_shapes(L, A, mid_genome):
    """Every place and size of a single indel against the oracle, at window geometries of 1-3 records per step,
    at the limit of the shifted windows (A + indel <= 248) and without the fast path (A + L > 248)."""
    batch = batch_from_records(_indel_records(mid_genome, 6000, 100 + L))
    libs = [("s", "a"), ("s", "b")]
    want = oracle_tableset(mid_genome, batch, libs, L, A, 0)
    got = run_engine(mid_genome, batch, libs, L, A, 0, resident=True)
    assert_tables_equal(got, want)


def _cigar_records(ref, specs, seed):
    """Records from (tid, pos, flag, [(op, len), ...]) with the read copied from the reference (5 % substitutions,
    random inserted / clipped bases)."""
    rng = np.random.default_rng(seed)
    bases, offs = ref.concat()
    upper = bases & np.uint8(0xDF)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    recs = []
    for tid, pos, flag, ops in specs:
        seq, r = [], offs[tid] + pos
        for op, ln in ops:
            if op in (0, 7, 8):
                seq.append(upper[r:r + ln].copy()); r += ln
            elif op in (1, 4):
                seq.append(rng.choice(acgt, ln))
            elif op in (2, 3):
                r += ln
        seq = np.concatenate(seq)
        seq = np.where(rng.random(seq.shape[0]) < 0.05, rng.choice(acgt, seq.shape[0]), seq)
        recs.append(dict(flag=flag, tid=tid, pos=pos, cigar=list(ops), seq=seq.tobytes().decode(), qual=None, lib=0,
                         tlen=0))
    return recs


@pytest.mark.parametrize("L,A", [(70, 10), (300, 20)])
def test_hip_long_reads_and_long_cigars(L, A, mid_genome):
    """Reads of 32 KiB and more (no 15-bit query length: the CIGAR walk), CIGARs of more than 64 operations (beyond
    the one-operation-per-lane preload), single-indel reads too long for the fast path, between ordinary reads."""
    many = []
    for k in range(60):
        many += [(0, 37), (1, 1 + k % 3), (0, 41), (2, 1 + k % 2)]
    many += [(0, 50)]
    specs = [(0, 1000, 0, [(0, 40000)]), (0, 50000, 16, [(0, 33000)]),
             (0, 100000, 0, [(0, 20000), (1, 5), (0, 20000)]), (0, 150000, 16, [(0, 17000), (2, 7), (0, 17000)]),
             (0, 200000, 0, [(4, 3)] + many + [(4, 2)]), (1, 20000, 16, many),
             (1, 40000, 0, [(0, 32767)]), (1, 1000, 0, [(0, 32768)]),
             (0, 250000, 0, [(0, 30), (3, 200), (0, 30), (1, 2), (0, 30)])]
    rng = np.random.default_rng(5)
    for i in range(300):   # ordinary records around them (tiles mix the kinds)
        specs.insert(int(rng.integers(0, len(specs) + 1)),
                     (int(rng.integers(0, 2)), int(rng.integers(300, 90000)), int(rng.choice([0, 16])),
                      [(0, int(rng.integers(1, 150)))]))
    batch = batch_from_records(_cigar_records(mid_genome, specs, 77))
    libs = [("s", "l")]
    want = oracle_tableset(mid_genome, batch, libs, L, A, 0)
    got = run_engine(mid_genome, batch, libs, L, A, 0, resident=True)
    assert_tables_equal(got, want)


@pytest.mark.parametrize("L,A", [(70, 10), (25, 4)])
def test_hip_single_indel_entries_of_phase_1_at_their_limits(L, A, mid_genome):
    """Records `M a, I|D g, M b` around the conditions under which the tile loop's phase 1 makes their single-indel entry itself
    (csrc/mdx_kernels.hip: SIP) instead of leaving them to the general pass: gaps of 7 and 8 bases, runs of one base, `=` / `X`
    as the match operations, flanks complete by one base and short by one at both ends of a contig, the first and the last
    records of the batch (the window loads' slack in the SEQ column), between plain records of both strands."""
    lens = [len(s) for s in mid_genome.seqs]
    specs = []
    for g in (1, 2, 7, 8):
        for a, b in ((1, 60), (60, 1), (1, 1), (5, 69), (69, 5), (70, 70), (71, 30), (30, 71), (120, 3)):
            for op in (1, 2):
                for flag in (0, 16):
                    for m_op in (0, 7, 8):
                        pos = 5000 + 37 * len(specs)
                        specs.append((0, pos, flag, [(m_op, a), (op, g), (m_op if m_op != 8 else 0, b)]))
    # flanks: complete by exactly A, short by one; at the contig's end the same
    for op in (1, 2):
        n0 = lambda a_, g_, b_: a_ + b_ + (g_ if op == 2 else 0)
        for tid in (0, 1):
            specs += [(tid, A, 0, [(0, 40), (op, 3), (0, 40)]), (tid, max(A - 1, 0), 16, [(0, 40), (op, 3), (0, 40)]),
                      (tid, lens[tid] - A - n0(40, 3, 40), 0, [(0, 40), (op, 3), (0, 40)]),
                      (tid, lens[tid] - A - n0(40, 3, 40) + 1, 16, [(0, 40), (op, 3), (0, 40)])]
    rng = np.random.default_rng(9)
    plain = [(int(rng.integers(0, 2)), int(rng.integers(300, 90000)), int(rng.choice([0, 16])), [(0, int(rng.integers(20, 150)))])
             for _ in range(400)]
    order = rng.permutation(len(specs) + len(plain))
    mixed = [(specs + plain)[i] for i in order]
    # (a single-indel record first and last in the batch: no slack in front of / behind its SEQ)
    mixed = [(0, 700, 0, [(0, 30), (1, 2), (0, 30)])] + mixed + [(1, 900, 16, [(0, 30), (2, 2), (0, 30)])]
    batch = batch_from_records(_cigar_records(mid_genome, mixed, 123))
    libs = [("s", "l")]
    want = oracle_tableset(mid_genome, batch, libs, L, A, 0)
    for resident in (True, False):
        got = run_engine(mid_genome, batch, libs, L, A, 0, resident=resident)
        assert_tables_equal(got, want)


@pytest.mark.parametrize("L,A", [(70, 10), (8, 3), (150, 30), (100, 12)])
def test_hip_single_indel_shapes_with_min_basequal(L, A, mid_genome):
    """The same shapes under --min-basequal 20: masked columns in the near and far entries, inserted columns that are
    masked, deleted columns (never masked), records without qualities."""
    batch = batch_from_records(_indel_records(mid_genome, 6000, 300 + L, with_qual=True), with_qual=True)
    libs = [("s", "a"), ("s", "b")]
    want = oracle_tableset(mid_genome, batch, libs, L, A, 20)
    got = run_engine(mid_genome, batch, libs, L, A, 20, resident=True)
    assert_tables_equal(got, want)


@pytest.mark.parametrize("k", range(10))
def test_hip_fuzzed_cigars_match_oracle(k, mid_genome):
    """Random CIGARs (M I D N P = X, clips, adjacent indels), flags, insert sizes, three libraries, qualities:
    the generator of tools/fuzz_vs_reference.py (there the oracle is held against the reference itself)."""
    from tools.fuzz_vs_reference import fuzz_records
    # (the last two: tables of one library too large for the LDS - global atomics, generic code)
    L, A = [(70, 10), (8, 3), (25, 5), (150, 30), (1, 0), (100, 12), (240, 8), (300, 10), (700, 10), (650, 40)][k]
    Q = [0, 20, 0, 15, 0, 30, 20, 0, 0, 20][k]
    batch = batch_from_records(fuzz_records(mid_genome, 4000, 7000 + k, with_qual=Q > 0), with_qual=True if Q > 0 else None)
    libs = [("Zed", "libB"), ("Alpha", "libA"), ("Mid", "libC")]
    want = oracle_tableset(mid_genome, batch, libs, L, A, Q)
    got = run_engine(mid_genome, batch, libs, L, A, Q)
    assert_tables_equal(got, want)


def test_hip_quality_hint_skips_nothing_that_matters(mid_genome):
    """MDX_FLAG_QUAL_ABOVE_MIN: records whose lowest quality is at or above --min-basequal carry the hint (their
    quality windows are not loaded); the tables must equal the oracle's, which never sees the bit.  Mixed batch:
    clean records, records with low bases, records without qualities."""
    from mapdamage_amd.batch import concat_batches, mark_unmaskable, record_min_quality
    from mapdamage_amd.engine import DamageEngine
    rng = np.random.default_rng(12)
    parts = []
    for k, clean in enumerate((True, False, True)):
        b = synth.make_reads(mid_genome, 30_000, 40 + k, len_range=(30, 140), paired=True, frac_softclip=0.1, frac_ins=0.05,
                             frac_del=0.05, frac_skip=0.01, with_qual=True, frac_filtered=0.02)
        if clean:
            b.qual = rng.integers(25, 42, b.qual.shape[0]).astype(np.uint8)
        parts.append(b)
    batch = concat_batches(parts)
    libs = [("s", "l")]
    want = oracle_tableset(mid_genome, batch, libs, 70, 10, 25)
    hinted, nothing = mark_unmaskable(batch, 25)
    assert not nothing and 0.5 < ((hinted.flag & 0x8000) != 0).mean() < 0.8
    assert (record_min_quality(batch)[(hinted.flag & 0x8000) != 0] >= 25).all()
    with DamageEngine(libs, 70, 10, 25) as eng:
        eng.set_reference(mid_genome)
        eng.tabulate(hinted)
        got = eng.finish()
    assert_tables_equal(got, want)


@pytest.mark.gpu
def test_hip_host_batches_in_flight_and_record_base():
    """mdx_tabulate_host returns once the host columns are staged: several slices are enqueued before the one sync (the
    copy of slice k+1 runs under the kernel of slice k, two staging sets), the tables equal the oracle's, and a bad
    record in a later slice is reported with its index among all records (mdx_set_record_base)."""
    from mapdamage_amd.engine import BadReadError, DamageEngine
    ref = synth.small_genome()
    batch = synth.make_reads(ref, 30_000, 5, len_range=(25, 140), nlib=2, paired=True, frac_softclip=0.1, frac_ins=0.05,
                             frac_del=0.05, frac_skip=0.01, frac_filtered=0.03)
    libs = [("a", "x"), ("b", "y")]
    want = oracle_tableset(ref, batch, libs, 70, 10, 0)
    cuts = [0, 7_000, 7_001, 16_000, 16_500, 29_999, 30_000]          # slices of very different sizes: the sets grow
    with DamageEngine(libs, 70, 10, 0) as eng:
        eng.set_reference(ref)
        for rep in range(3):
            eng.reset()
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                eng.tabulate(batch.slice(lo, hi), sync=False, record_base=lo)
            assert_tables_equal(eng.finish(), want)
        # a record past its contig end in the fourth slice
        eng.reset()
        bad_at = int(np.flatnonzero((batch.flag[16_000:16_500] & 0xF04) == 0)[7]) + 16_000
        broken = batch.slice(0, batch.n)
        broken.pos[bad_at] = 10_000_000
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            eng.tabulate(broken.slice(lo, hi), sync=False, record_base=lo)
        with pytest.raises(BadReadError) as err:
            eng.sync()
        assert err.value.read_index == bad_at
        # and the default call still synchronises and numbers within the batch
        eng.reset()
        with pytest.raises(BadReadError) as err:
            eng.tabulate(broken.slice(16_000, 16_500))
        assert err.value.read_index == bad_at - 16_000


@pytest.mark.gpu
def test_hip_a_batch_ordered_by_library_belongs_to_the_layout_it_was_built_for(mid_genome):
    """mdx_batch::libsort (the resident batch's copy ordered by library, statistics.py:12-20) is laid out for the batch's sizes
    and the libraries of the context that uploaded it: a context with another number of libraries, or a view of a part of the
    batch, refuses it instead of reading it through the wrong offsets."""
    import ctypes

    from mapdamage_amd.engine import DamageEngine, MdxBatch, MdxError
    batch = synth.make_reads(mid_genome, 20_000, 23, len_range=(30, 120), nlib=2, paired=True, frac_softclip=0.1, frac_ins=0.04,
                             frac_del=0.04)
    libs2, libs3 = [("s", "a"), ("s", "b")], [("s", "a"), ("s", "b"), ("s", "c")]
    with DamageEngine(libs2, 70, 10, 0) as eng2, DamageEngine(libs3, 70, 10, 0) as eng3:
        eng2.set_reference(mid_genome)
        eng3.set_reference(mid_genome)
        db = eng2.upload(batch, packed=True)
        assert db.dev.libsort
        eng2.tabulate(db)
        assert_tables_equal(eng2.finish(), oracle_tableset(mid_genome, batch, libs2, 70, 10, 0))
        view = MdxBatch()
        ctypes.memmove(ctypes.byref(view), ctypes.byref(db.dev), ctypes.sizeof(MdxBatch))
        with pytest.raises(MdxError, match="mdx_batch::libsort was built for another batch"):
            eng3.tabulate_view(view)
        view.n_reads = batch.n // 2
        view.n_cigar = int(batch.cigar_off[batch.n // 2])
        view.n_bases = int(batch.seq_off[batch.n // 2])
        with pytest.raises(MdxError, match="mdx_batch::libsort was built for another batch"):
            eng2.tabulate_view(view)
        # (without the copy the same views are sorted inside the launch and counted)
        view.libsort = None
        eng2.reset()
        eng2.tabulate_view(view)
        assert_tables_equal(eng2.finish(), oracle_tableset(mid_genome, batch.slice(0, batch.n // 2), libs2, 70, 10, 0))
        db.free()
