"""Worker of tests/test_gpu_distributed.py (launched under torch.distributed.run, one rank per GPU, RCCL).

One batch is sharded over the ranks with shard_bounds; every rank tabulates its shard on its own GPU with the
HIP engine, and the tables are reduced by both routes:
  torch  — mapdamage_amd.distributed.reduce_engine_tables (torch.distributed all-reduce + tensor gathers);
  capi   — the library's own RCCL communicator (mdx_comm_init, collective mdx_finish).
Every rank compares the totals with the oracle over the whole batch (tests/ may use the oracle).  Last: a bad
record on one rank must raise on every rank, not hang."""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.batch import concat_batches  # noqa: E402
from mapdamage_amd.distributed import adopt_torch_rccl, attach_rccl, reduce_engine_tables, shard_bounds  # noqa: E402
from mapdamage_amd.engine import BadReadError, DamageEngine, MdxError  # noqa: E402
from tests.util import assert_tables_equal, oracle_tableset  # noqa: E402


def main():
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", local)
    ref, small = synth.config1_batch()
    big = synth.make_reads(ref, 40_000, 77, len_range=(30, 120), nlib=2, paired=True, frac_softclip=0.1,
                           frac_ins=0.05, frac_del=0.05, frac_skip=0.01, frac_filtered=0.03, with_qual=True)
    batch = concat_batches([small, big, small])     # out-of-range lengths (tlen = 70000) at both ends
    libs = [("Zed", "libB"), ("Alpha", "libA")]
    L, A, lgd_max = 70, 10, 4096
    lo, hi = shard_bounds(batch.n, rank, world)
    for Q in (0, 20):
        want = oracle_tableset(ref, batch, libs, L, A, Q, lgd_max)
        with DamageEngine(libs, L, A, Q, lgd_max=lgd_max, device=local) as eng:
            eng.set_reference(ref)
            eng.tabulate(batch.slice(lo, hi))
            total = reduce_engine_tables(eng, dev)
            assert_tables_equal(total, want)
            assert total.misincorporation_text() == want.misincorporation_text()
            assert total.lgdistribution_text() == want.lgdistribution_text()
            # the same totals through the library's own communicator (collective finish)
            assert eng.comm_count() == 0
            attach_rccl(eng)
            assert eng.comm_size == world and eng.comm_count() == world     # (ncclCommCount of the library's own communicator)
            total2 = eng.finish()
            assert_tables_equal(total2, want)
            assert total2.lgdistribution_text() == want.lgdistribution_text()
            # ... and as a device buffer
            words = torch.empty(eng.table_words(), dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            eng.finish_allreduce(words.data_ptr())
            eng.sync()
            total3 = eng.unpack_tables(words.cpu().numpy().view(np.uint64), total2.lgd_over)
            assert_tables_equal(total3, want)

    # the C-ABI reduction over the communicator torch itself uses (mdx_comm_adopt)
    want = oracle_tableset(ref, batch, libs, L, A, 0, lgd_max)
    with DamageEngine(libs, L, A, 0, lgd_max=lgd_max, device=local) as eng:
        eng.set_reference(ref)
        eng.tabulate(batch.slice(lo, hi))
        if adopt_torch_rccl(eng, dev):
            assert eng.comm_size == world and eng.comm_count() == world
            assert_tables_equal(eng.finish(), want)
            adopted = "adopted"
        else:
            adopted = "no _comm_ptr in this torch"

    # a record past its contig end on the last rank: every rank raises
    with DamageEngine(libs, L, A, 0, lgd_max=lgd_max, device=local) as eng:
        eng.set_reference(ref)
        mine = batch.slice(lo, hi)
        if rank == world - 1:
            mine.pos = mine.pos.copy()
            mine.pos[3] = 10_000_000
        dbatch = eng.upload(mine)
        eng.tabulate(dbatch)
        try:
            reduce_engine_tables(eng, dev)
            outcome = "none"
        except BadReadError as exc:
            outcome = "own" if exc.read_index == 3 else "own-wrong-index"
        except RuntimeError:
            outcome = "peer"
        assert outcome == ("own" if rank == world - 1 else "peer"), outcome
        attach_rccl(eng)
        try:
            eng.finish()
            outcome = "none"
        except (BadReadError, MdxError) as exc:
            outcome = "own" if getattr(exc, "code", -6) == -6 else "peer"
        assert outcome == ("own" if rank == world - 1 else "peer"), outcome
        dbatch.free()
    dist.barrier()
    if rank == 0:
        print("gpu dist ok: world=%d kept=%d torch communicator: %s" % (world, total.n_kept, adopted))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
