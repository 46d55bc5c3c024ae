"""Worker for the world_size-2 gloo test: shard-by-read + table all-reduce (CPU only).

Each rank tabulates its shard with the C oracle (this file is under tests/: the oracle is
allowed here), packs the tables into the engine's canonical word block (mapdamage_amd/layout.py — the
very message the GPU path all-reduces), all-reduces it with gloo, gathers the out-of-range length
lists with the fixed-size tensor collectives, and checks the result against the oracle run over the
whole batch.  Then the error agreement: one rank fails, every rank must raise instead of hanging."""

import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.distributed import agree_on_error, gather_lgd_overflow, reduce_tableset, shard_bounds  # noqa: E402
from tests.util import assert_tables_equal, oracle_tableset  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ref, batch = synth.config1_batch()
    libs = [("Zed", "libB"), ("Alpha", "libA")]
    L, A, Q, lgd_max = 70, 10, 20, 4096
    lo, hi = shard_bounds(batch.n, rank, world)
    mine = oracle_tableset(ref, batch.slice(lo, hi), libs, L, A, Q, lgd_max)
    total = reduce_tableset(mine, lgd_max)
    want = oracle_tableset(ref, batch, libs, L, A, Q, lgd_max)
    assert_tables_equal(total, want)
    assert total.misincorporation_text() == want.misincorporation_text()
    assert total.lgdistribution_text() == want.lgdistribution_text()
    assert len(total.lgd_over) == len(want.lgd_over) > 0  # the tlen=70000 records travel by all_gather
    # ragged lists, rank order, an empty one among them
    own = np.arange(4 * (3 * rank), dtype=np.int64).reshape(-1, 4) + 1000 * rank
    everyone = gather_lgd_overflow(own)
    expect = np.concatenate([np.arange(4 * (3 * r), dtype=np.int64).reshape(-1, 4) + 1000 * r for r in range(world)])
    assert np.array_equal(everyone, expect), (everyone, expect)
    assert gather_lgd_overflow(np.zeros((0, 4), np.int64)).shape == (0, 4)
    # error agreement: the last rank fails, everybody raises, nobody waits in a collective
    try:
        agree_on_error(ValueError("bad record on rank %d" % rank) if rank == world - 1 else None)
        raised = None
    except ValueError as exc:
        raised = "own:" + str(exc)
    except RuntimeError as exc:
        raised = "peer:" + str(exc)
    assert raised is not None and raised.startswith("own:" if rank == world - 1 else "peer:"), raised
    agree_on_error(None)   # and no error: no exception
    dist.barrier()
    if rank == 0:
        print("dist ok: world=%d kept=%d" % (world, total.n_kept))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
