"""Pins the C oracle (oracle/mdx_oracle.c) to the golden vectors produced by the reference's
own Python functions (tools/make_golden.py).  CPU only."""

import numpy as np
import pytest

from tests.util import Golden, golden_names, oracle_tableset


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_golden(name):
    g = Golden(name)
    # lgd_max small enough that the tlen=70000 record exercises the overflow list
    ts = oracle_tableset(g.ref, g.batch, g.libraries, g.length, g.around, g.minqual, lgd_max=4096)
    g.check(ts)


def test_oracle_rejects_alignment_past_contig_end():
    """FastaFile.fetch(start > end) raises ValueError in the reference (align.py:33)."""
    from mapdamage_amd import synth
    from mapdamage_amd.batch import batch_from_records
    from oracle.oracle import OracleError, tabulate
    ref = synth.small_genome()
    n_last = ref.lengths[-1]
    recs = [dict(flag=0, tid=0, pos=10, cigar=[(0, 20)], seq="A" * 20, qual=None, lib=0, tlen=0),
            dict(flag=0, tid=2, pos=n_last - 10, cigar=[(0, 25)], seq="A" * 25, qual=None, lib=0,
                 tlen=0)]
    with pytest.raises(OracleError) as e:
        tabulate(ref, batch_from_records(recs), 1, 70, 10)
    assert e.value.code == -6 and e.value.read_index == 1
    # ending exactly at the contig end is fine (empty 'after' flank)
    recs[1]["pos"] = n_last - 25
    out = tabulate(ref, batch_from_records(recs), 1, 70, 10)
    assert out["n_kept"] == 2


def test_golden_fixture_inventory():
    names = golden_names()
    assert len(names) >= 15 and not any(n.startswith("genome_") for n in names)
    for n in names:
        g = Golden(n)
        assert g.batch.n == g.meta["n_reads"]
        assert g.mis.shape[1:] == (2, 2, g.length, 25)


def test_parallel_oracle_equals_serial_oracle():
    """bench.py's all-cores CPU baseline (oracle.tabulate_parallel) sums per-slice tables."""
    import numpy as np
    from mapdamage_amd import synth
    from oracle import oracle
    ref = synth.small_genome()
    batch = synth.make_reads(ref, 4000, 5, len_range=(20, 90), frac_softclip=0.2, frac_ins=0.1, frac_del=0.1,
                             paired=True, nlib=2)
    one = oracle.tabulate(ref, batch, 2, 25, 4, 0, lgd_max=512)
    many, n = oracle.tabulate_parallel(ref, batch, 2, 25, 4, 0, lgd_max=512, threads=3)
    assert n == 3 and many["n_kept"] == one["n_kept"]
    for k in ("mis", "comp", "lgd"):
        np.testing.assert_array_equal(many[k], one[k])
