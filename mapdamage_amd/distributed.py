"""Shard-by-read across the GPUs of one node (one process per GPU) + one all-reduce of the tables.

The path shards trivially: every record contributes independent integer increments
(statistics.py:30,35,40,103,124,126 are all ``+= 1``), so rank r tabulates records
[n*r/W, n*(r+1)/W) with no data-path exchange, and the only collective is a SUM all-reduce of
the packed uint64 table block at the end of the pass (RCCL over xGMI with backend "nccl";
"gloo" on CPU for the tests).  The message is ~66 KB per library plus the dense length
histogram, i.e. latency-bound: a single all-reduce, no bucketing.

Two equivalent routes:
* ``reduce_engine_tables`` — through torch.distributed (``bench.py --gpus N`` and ``python -m mapdamage_amd
  --gpus N`` use it: the process group exists anyway; main.py ``_Ranks``);
* the C-ABI route (``DamageEngine.comm_init`` + ``finish()``, include/mdx.h ``mdx_comm_*``): the library
  calls RCCL itself, for consumers without torch.  ``attach_rccl`` wires it up from a torch process group
  (the unique id travels through ``broadcast_object_list``).
"""

import numpy as np

from .tables import TableSet, unpack_words


def shard_bounds(n, rank, world):
    return (n * rank) // world, (n * (rank + 1)) // world


def _active():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def allreduce_words(words):
    """In-place SUM all-reduce of a packed table block held in a torch int64 tensor (CUDA for
    RCCL, CPU for gloo).  uint64 counters are summed as two's-complement int64: identical bits."""
    import torch.distributed as dist
    if _active():
        dist.all_reduce(words, op=dist.ReduceOp.SUM)
    return words


def gather_lgd_overflow(over, device=None):
    """Concatenate every rank's out-of-range fragment-length records (rare, variable length) in rank
    order: the lengths are all-gathered first, then the lists padded to the longest — two fixed-size
    tensor collectives, no pickling."""
    import torch
    import torch.distributed as dist
    over = np.ascontiguousarray(np.asarray(over, dtype=np.int64).reshape(-1, 4))
    if not _active() or dist.get_world_size() == 1:
        return over
    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    mine = torch.tensor([over.shape[0]], dtype=torch.int64, device=dev)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine)
    counts = counts.cpu().numpy()
    longest = int(counts.max())
    if longest == 0:
        return over
    padded = torch.zeros((longest, 4), dtype=torch.int64, device=dev)
    if over.shape[0]:
        padded[:over.shape[0]] = torch.from_numpy(over).to(dev)
    everyone = torch.zeros((world * longest, 4), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(everyone, padded)
    everyone = everyone.cpu().numpy().reshape(world, longest, 4)
    return np.concatenate([everyone[r, :int(counts[r])] for r in range(world)])


def agree_on_error(error, device=None):
    """Every rank learns whether any rank failed (MAX all-reduce of a flag) *before* the table collectives, so
    that a rank whose batch held a bad record cannot leave the others waiting in the all-reduce.  Re-raises the
    rank's own error; raises RuntimeError on the healthy ranks."""
    import torch
    import torch.distributed as dist
    if _active() and dist.get_world_size() > 1:
        flag = torch.tensor([1 if error is not None else 0], dtype=torch.int64, device=device if device is not None else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if error is None and int(flag.item()):
            raise RuntimeError("another rank failed while tabulating; no tables were reduced")
    if error is not None:
        raise error


def reduce_engine_tables(engine, device) -> TableSet:
    """sync -> error agreement -> finish_device -> all-reduce -> TableSet on every rank (GPU path).
    The engine runs on its own HIP stream: torch's stream is drained before the engine writes the buffer
    torch allocated, and the engine's before torch reads it."""
    import torch
    error = None
    try:
        engine.sync()
    except Exception as exc:          # BadReadError / MdxError of this rank
        error = exc
    agree_on_error(error, device)
    words = torch.empty(engine.table_words(), dtype=torch.int64, device=device)
    torch.cuda.synchronize(device)
    engine.finish_device(words.data_ptr())
    engine.sync()
    allreduce_words(words)
    over = gather_lgd_overflow(engine.lgd_overflow_only(), device)
    host = words.cpu().numpy().view(np.uint64)
    return unpack_words(host, engine.libraries, engine.length, engine.around, engine.lgd_max, over)


def attach_rccl(engine):
    """Give the engine its own RCCL communicator over the ranks of torch's default process group (the C-ABI
    route): afterwards ``engine.finish()`` is collective and returns the totals on every rank."""
    import torch.distributed as dist
    world, rank = (dist.get_world_size(), dist.get_rank()) if _active() else (1, 0)
    box = [engine.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    engine.comm_init(box[0], world, rank)
    return engine


def adopt_torch_rccl(engine, device):
    """Hand the engine the ``ncclComm_t`` torch's default process group uses on ``device`` (mdx_comm_adopt): the C-ABI
    reduction then runs over the very communicator torch.distributed has set up.  Returns False when this torch build
    does not expose the pointer (``ProcessGroupNCCL._comm_ptr``)."""
    import torch
    import torch.distributed as dist
    if not _active() or dist.get_backend() != "nccl":
        return False
    # (the communicator exists once a collective has run on the device)
    dist.all_reduce(torch.zeros(1, dtype=torch.int64, device=device))
    torch.cuda.synchronize(device)
    pg = dist.distributed_c10d._get_default_group()._get_backend(torch.device(device))
    ptr = getattr(pg, "_comm_ptr", None)
    if ptr is None:
        return False
    engine.comm_adopt(int(ptr()), dist.get_world_size(), dist.get_rank())
    return True


def reduce_tableset(ts: TableSet, lgd_max) -> TableSet:
    """All-reduce a host TableSet (CPU/gloo path used by the tests)."""
    import torch
    from .tables import pack_words
    words = torch.from_numpy(pack_words(ts).view(np.int64).copy())
    allreduce_words(words)
    over = gather_lgd_overflow(ts.lgd_over)
    return unpack_words(words.numpy().view(np.uint64), ts.libraries, ts.length, ts.around, lgd_max, over)
