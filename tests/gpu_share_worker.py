"""Worker of tests/test_gpu_distributed.py::test_two_engines_on_one_gpu_*: N >= 2 ranks on ONE GPU (every rank
opens cuda:0), so that shard -> HIP engine -> packed block -> reduce -> unpack runs with real device tables on a
1-GPU box.  RCCL refuses two ranks on one device, so the blocks are summed over gloo (host tensors) — the same
message, the same reduce_tableset / gather_lgd_overflow / agree_on_error code the RCCL route shares.

The loop being sharded is mapdamage/main.py:165-217; the sum over shards is legal because every accumulator update
is `+= 1` (statistics.py:30,35,40,103,124,126)."""

import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.batch import concat_batches  # noqa: E402
from mapdamage_amd.distributed import agree_on_error, reduce_tableset, shard_bounds  # noqa: E402
from mapdamage_amd.engine import BadReadError, DamageEngine  # noqa: E402
from mapdamage_amd.tables import pack_words  # noqa: E402
from tests.util import assert_tables_equal, oracle_tableset  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ref, small = synth.config1_batch()
    big = synth.make_reads(ref, 60_000, 78, len_range=(30, 120), nlib=2, paired=True, frac_softclip=0.1,
                           frac_ins=0.05, frac_del=0.05, frac_skip=0.01, frac_filtered=0.03, with_qual=True)
    batch = concat_batches([small, big, small])     # out-of-range fragment lengths (tlen = 70000) in both end shards
    libs = [("Zed", "libB"), ("Alpha", "libA")]
    L, A, lgd_max = 70, 10, 4096
    lo, hi = shard_bounds(batch.n, rank, world)
    for Q in (0, 20):
        want = oracle_tableset(ref, batch, libs, L, A, Q, lgd_max)
        with DamageEngine(libs, L, A, Q, lgd_max=lgd_max, device=0) as eng:
            eng.set_reference(ref)
            dbt = eng.upload(batch.slice(lo, hi))
            eng.tabulate(dbt)
            mine = eng.finish()
            dbt.free()
            # this rank's own block against the oracle over its own shard ...
            assert_tables_equal(mine, oracle_tableset(ref, batch.slice(lo, hi), libs, L, A, Q, lgd_max))
            # ... the reduced block against the oracle over the whole batch, and against one engine that counts it all
            total = reduce_tableset(mine, lgd_max)
            assert_tables_equal(total, want)
            assert total.misincorporation_text() == want.misincorporation_text()
            assert total.dnacomp_text() == want.dnacomp_text()
            assert total.lgdistribution_text() == want.lgdistribution_text()
            assert len(total.lgd_over) == len(want.lgd_over) > 0
            eng.reset()
            eng.tabulate(batch)
            single = eng.finish()
            assert np.array_equal(pack_words(single)[:-1], pack_words(total)[:-1])

    # a record past its contig end on rank 1: its own BadReadError there, RuntimeError on the others, nobody waits
    with DamageEngine(libs, L, A, 0, lgd_max=lgd_max, device=0) as eng:
        eng.set_reference(ref)
        mine = batch.slice(lo, hi)
        bad_at = int(np.flatnonzero((mine.flag & 0xF04) == 0)[5])      # a record the flag filter keeps
        if rank == 1:
            mine.pos = mine.pos.copy()
            mine.pos[bad_at] = 10_000_000
        eng.tabulate(eng.upload(mine))
        error = None
        try:
            eng.sync()
        except BadReadError as exc:
            error = exc
        try:
            agree_on_error(error)
            outcome = "none"
        except BadReadError as exc:
            outcome = "own" if exc.read_index == bad_at else "own-wrong-index"
        except RuntimeError:
            outcome = "peer"
        assert outcome == ("own" if rank == 1 else "peer"), outcome
    dist.barrier()
    if rank == 0:
        print("gpu share ok: world=%d kept=%d" % (world, total.n_kept))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
