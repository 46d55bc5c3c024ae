"""Shard-by-read across the GPUs of one node (one process per GPU) + one all-reduce of the tables.

The path shards trivially: every record contributes independent integer increments
(statistics.py:30,35,40,103,124,126 are all ``+= 1``), so rank r tabulates records
[n*r/W, n*(r+1)/W) with no data-path exchange, and the only collective is a SUM all-reduce of
the packed uint64 table block at the end of the pass (RCCL over xGMI with backend "nccl";
"gloo" on CPU for the tests).  The message is ~66 KB per library plus the dense length
histogram, i.e. latency-bound: a single all-reduce, no bucketing.
"""

import numpy as np

from .tables import TableSet, unpack_words


def shard_bounds(n, rank, world):
    return (n * rank) // world, (n * (rank + 1)) // world


def allreduce_words(words):
    """In-place SUM all-reduce of a packed table block held in a torch int64 tensor (CUDA for
    RCCL, CPU for gloo).  uint64 counters are summed as two's-complement int64: identical bits."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(words, op=dist.ReduceOp.SUM)
    return words


def gather_lgd_overflow(over):
    """Concatenate every rank's out-of-range fragment-length records (rare, variable length)."""
    import torch.distributed as dist
    over = np.asarray(over, dtype=np.int64).reshape(-1, 4)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return over
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, over)
    return np.concatenate(parts) if parts else over


def reduce_engine_tables(engine, device):
    """finish_device -> all-reduce -> TableSet on every rank (GPU path)."""
    import torch
    words = torch.zeros(engine.table_words(), dtype=torch.int64, device=device)
    engine.finish_device(words.data_ptr())
    engine.sync()
    allreduce_words(words)
    over = gather_lgd_overflow(engine.lgd_overflow_only())
    host = words.cpu().numpy().view(np.uint64)
    return unpack_words(host, engine.libraries, engine.length, engine.around, engine.lgd_max, over)


def reduce_tableset(ts: TableSet, lgd_max) -> TableSet:
    """All-reduce a host TableSet (CPU/gloo path used by the tests)."""
    import torch
    from .tables import pack_words
    words = torch.from_numpy(pack_words(ts).view(np.int64).copy())
    allreduce_words(words)
    over = gather_lgd_overflow(ts.lgd_over)
    return unpack_words(words.numpy().view(np.uint64), ts.libraries, ts.length, ts.around, lgd_max, over)
