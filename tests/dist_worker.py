"""Worker for the world_size-2 gloo test: shard-by-read + table all-reduce (CPU only).

Each rank tabulates its shard with the C oracle (this file is under tests/: the oracle is
allowed here), all-reduces the packed table block with gloo, and checks the result against the
oracle run over the whole batch."""

import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.distributed import reduce_tableset, shard_bounds  # noqa: E402
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
    dist.barrier()
    if rank == 0:
        print("dist ok: world=%d kept=%d" % (world, total.n_kept))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
