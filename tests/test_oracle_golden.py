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
