"""Mirror of the reference's command line for the tabulation pass (mapdamage/main.py:49-266,
mapdamage/config.py:80-494): same flags, same defaults, same output files
(``misincorporation.txt``, ``dnacomp.txt``, ``lgdistribution.txt``, ``Runtime_log.txt``), with the
per-read loop replaced by ``DamageEngine`` (HIP).  R plotting, the Bayesian stage and rescaling
are out of scope: their flags are parsed, and asking for them is an error."""

import argparse
import dataclasses
import logging
import sys
import time
from pathlib import Path

import numpy as np

from . import __version__
from .batch import mark_unmaskable
from .engine import BadReadError, DamageEngine, MdxError
from .fasta import compare_sequence_dicts, read_fasta_index, reference_for_bam
from .layout import FLAG_FILTER
from .reader import BAMReader, draw_uniform
from .sam import BAMError
from .statistics import check_table_and_warn_if_dmg_freq_is_low

_LOG_FORMAT = "%(asctime)s %(name)s %(levelname)s %(message)s"


class _Stages:
    """MDX_STAGE_LOG=<file>: the wall-clock time (``time.time()``) at which each stage of the run was reached, as one JSON
    object — what bench.py's ``cli_wall`` splits a cold run of this command into.  Costs a dictionary entry per stage."""

    def __init__(self):
        import os
        self.path = os.environ.get("MDX_STAGE_LOG")
        self.marks = []

    def mark(self, label):
        if self.path:
            self.marks.append((label, time.time()))

    def write(self):
        if self.path:
            import json
            with open(self.path, "w") as fh:
                json.dump({"stages": self.marks}, fh)


def _warm_up(device, pinned_bytes):
    """On a helper thread beside the header and index reads: the device's context, the decode kernels' code object, the
    inflating threads and the pinned buffer of the host's share (include/mdx.h ``mdx_warm``; ctypes drops the GIL)."""
    from .engine import load_library
    try:
        load_library().mdx_warm(device, pinned_bytes)
    except Exception:       # (a run that cannot warm up finds out why when it creates its engine)
        pass


def _ranged(cls, lo=float("-inf"), hi=float("inf")):
    def parse(value):
        value = cls(value)
        if value < lo:
            raise argparse.ArgumentTypeError("must be greater than or equal to %s" % (lo,))
        if value > hi:
            raise argparse.ArgumentTypeError("must be less than or equal to %s" % (hi,))
        return value
    return parse


def build_parser():
    p = argparse.ArgumentParser(prog="mapDamage", usage="%(prog)s [options] -i alignment.bam -r reference.fasta")
    p.add_argument("--version", action="version", version="%(prog)s (mapdamage_amd " + __version__ + ")")
    g = p.add_argument_group("Input and output")
    g.add_argument("-i", "--input", dest="filename", type=Path, metavar="SAM/BAM")
    g.add_argument("-r", "--reference", dest="ref", type=Path, metavar="FASTA")
    g.add_argument("-d", "--folder", type=Path)
    g.add_argument("-n", "--downsample", type=float, metavar="X")
    g.add_argument("--downsample-seed", type=int, metavar="X")
    g = p.add_argument_group("General options")
    g.add_argument("--merge-libraries", action="store_true")
    g.add_argument("--merge-reference-sequences", action="store_true", help=argparse.SUPPRESS)
    g.add_argument("-l", "--length", type=_ranged(int, 1), default=70)
    g.add_argument("-a", "--around", type=_ranged(int, 0), default=10)
    g.add_argument("-Q", "--min-basequal", dest="minqual", type=_ranged(int, 0, 93), default=0)
    g.add_argument("--plot-only", action="store_true")
    g.add_argument("--log-level", default="INFO", type=str.upper, choices=("DEBUG", "INFO", "WARNING", "ERROR"))
    g.add_argument("--no-plot", dest="no_r", action="store_true", help=argparse.SUPPRESS)
    g = p.add_argument_group("Options for graphics")
    g.add_argument("-y", "--ymax", type=float, default=0.3)
    g.add_argument("-m", "--readplot", type=_ranged(int, 1), default=25)
    g.add_argument("-b", "--refplot", type=_ranged(int, 1), default=10)
    g.add_argument("-t", "--title")
    g = p.add_argument_group("Options for the statistical estimation")
    for flag, typ, default in (("--rand", int, 30), ("--burn", int, 10000), ("--adjust", int, 10),
                               ("--iter", int, 50000), ("--seq-length", int, 12)):
        g.add_argument(flag, type=typ, default=default)
    g.add_argument("--termini", choices=("5p", "3p", "both"), default="both")
    for flag in ("--forward", "--reverse", "--var-disp", "--jukes-cantor", "--diff-hangs", "--fix-nicks",
                 "--use-raw-nick-freq", "--single-stranded", "--theme-bw", "--stats-only", "--no-stats",
                 "--check-R-packages"):
        g.add_argument(flag, action="store_true")
    g = p.add_argument_group("Options for rescaling of BAM files")
    g.add_argument("--rescale", action="store_true")
    g.add_argument("--rescale-only", action="store_true")
    g.add_argument("--rescale-out", type=Path)
    g.add_argument("--rescale-length-5p", type=int)
    g.add_argument("--rescale-length-3p", type=int)
    g = p.add_argument_group("MI355X engine")
    g.add_argument("--device", type=int, default=0, help="HIP device ordinal")
    g.add_argument("--gpus", type=_ranged(int, 1), default=1,
                   help="tabulate on this many GPUs of the node: one process per GPU (the command re-executes itself under "
                        "torch.distributed.run unless it was launched that way), the records sharded by slab of the file, the "
                        "tables summed with one RCCL all-reduce, rank 0 writes the output files")
    g.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                   help="torch.distributed backend of the table reduction: nccl = RCCL over xGMI; gloo sums on the host")
    g.add_argument("--share-gpu", action="store_true",
                   help="every rank uses --device instead of its own GPU (tests of the multi-rank path on a 1-GPU box; "
                        "needs --dist-backend gloo: RCCL refuses two ranks on one device)")
    g.add_argument("--print-launch", action="store_true",
                   help="with --gpus N > 1: print the torchrun command the run would re-execute itself under, and exit")
    g.add_argument("--freq-files", action="store_true",
                   help="EXPERIMENTAL: also write 5pCtoT_freq.txt / 3pGtoA_freq.txt (mapDamage 2.0-2.2 outputs that "
                        "this reference snapshot no longer produces; format unpinned)")
    g.add_argument("--batch-reads", type=int, default=4_000_000, help="records per device batch")
    g.add_argument("--gpu-decode", dest="gpu_decode", action="store_true", default=True,
                   help="inflate and unpack a BAM file on the GPU (include/mdx.h mdx_gbam_*): the compressed file goes "
                        "to HBM, the batch columns never exist on the host (the default; SAM input and --downsample to a fixed "
                        "number of reads are decoded on the host)")
    g.add_argument("--host-decode", dest="gpu_decode", action="store_false",
                   help="decode on the host (multi-threaded BGZF/BAM decoder) even where the GPU path applies")
    g.add_argument("--host-deflate", action="store_true",
                   help="--rescale-only: deflate the output's BGZF blocks with zlib (level 6) on the host's threads, as htslib does "
                        "behind the reference, instead of on the device (the default: four times as fast, the file 3 %% larger; the "
                        "records are the same)")
    g.add_argument("--chunk-mb", type=_ranged(float, 0), default=1024,
                   help="decode a BAM file in chunks of this many MiB of uncompressed records, overlapped with "
                        "the tabulation of the previous chunk (0: decode the whole file first)")
    return p


def parse_args(argv):
    parser = build_parser()
    o = parser.parse_args(argv)
    if o.plot_only or o.stats_only or o.rescale or o.check_R_packages:
        parser.error("plotting and the Bayesian estimation are not part of this engine; run them with the "
                     "reference on the emitted tables (--rescale-only works from an existing "
                     "Stats_out_MCMC_correct_prob.csv)")
    if o.rescale_only and not o.folder:
        parser.error("--folder required when using --rescale-only")
    if not o.filename:
        parser.error("--input SAM/BAM file not specified")
    if not o.ref:
        parser.error("--reference FASTA file not specified")
    if o.downsample is not None:
        if o.downsample <= 0:
            parser.error("-n/--downsample must be a positive value")
        elif o.downsample >= 1:
            o.downsample = int(o.downsample)
    if o.ymax <= 0 or o.ymax > 1:
        parser.error("--ymax (-b) must be an real number beetween 0 and 1")
    if o.refplot > o.around:
        parser.error("--refplot (-b) must be less than --around (-a)")
    if o.readplot > o.length:
        parser.error("--readplot (-m) must be less than --length (-l)")
    if not o.folder:
        o.folder = Path(o.filename.stem + ".mapDamage")
    o.folder.mkdir(parents=True, exist_ok=True, mode=0o750)
    o.no_stats = True
    if not o.rescale_out and o.rescale_only:
        o.rescale_out = o.folder / (o.filename.stem + ".rescaled.bam")
    if o.rescale_length_3p is None:
        o.rescale_length_3p = o.seq_length
    elif not (0 <= o.rescale_length_3p <= o.seq_length):
        parser.error("--rescale-length-3p must be less than or equal to --seq-length and greater than zero")
    if o.rescale_length_5p is None:
        o.rescale_length_5p = o.seq_length
    elif not (0 <= o.rescale_length_5p <= o.seq_length):
        parser.error("--rescale-length-5p must be less than or equal to --seq-length and greater than zero")
    return o


def rescale_qual(options):
    """Mirror of rescale.rescale_qual (mapdamage/rescale.py:368-383) for --rescale-only."""
    from .rescale import RescaleError, RescaleModel, rescale_bam
    from .sam import BamStream
    logger = logging.getLogger(__name__)
    logger.info("Rescaling BAM: '%s' -> '%s'", options.filename, options.rescale_out)
    start = time.time()
    try:
        model = RescaleModel.from_csv(options.folder / "Stats_out_MCMC_correct_prob.csv",
                                      options.rescale_length_5p, options.rescale_length_3p)
        # the header only (the records are streamed by rescale_bam).  The reference's --rescale-only branch
        # (main.py:121-124) goes straight to rescale_qual without the .fai / dictionary checks of the tabulation
        # pass: a sequence the FASTA lacks, or holds at another length, only matters when a record maps there
        # (fetch fails at that read) — here such a record is a bad record when the kernel meets it.
        with BamStream(options.filename) as probe:
            header = probe.header
        ref = reference_for_bam(options.ref, header.references, missing_ok=True)
        for name, length, have in zip(header.references, header.lengths, ref.lengths):
            if have != length:
                logger.warning("FASTA sequence %r is %s; the BAM header says %i bp — records mapped to it may fail",
                               name, "missing" if not have else "%i bp" % have, length)
        with DamageEngine([("*", "*")], options.length, options.around, 0, device=options.device) as engine:
            summary = None
            if options.gpu_decode and not options.host_deflate and _device_path_applies(options):
                # the records never on the host: inflated, rescaled, written back and deflated in HBM
                from .rescale import rescale_bam_on_device
                from .sam import GpuDecodeUnsupported
                try:
                    summary, counts = rescale_bam_on_device(engine, ref, options.filename, options.rescale_out, model)
                except (GpuDecodeUnsupported, ValueError, MdxError) as error:
                    # (never silent; the host decoder reads the whole file again and words the errors as the reference does)
                    logger.warning("GPU decode path gave up: %s; rescaling through the host decoder (the whole file again)", error)
                    engine.reset()
                    summary = None
            if summary is None:
                summary, counts = rescale_bam(engine, ref, options.filename, options.rescale_out, model,
                                              device_deflate=not options.host_deflate)
    except RescaleError as error:
        logger.error("%s", error)
        return 1
    if counts["inward_pairs"] or counts["improper_pairs"]:
        logger.warning("Processed %i paired reads, assumed to be non-overlapping, facing inwards and correctly "
                       "paired; %i of these were excluded as improperly paired.",
                       counts["inward_pairs"] + counts["improper_pairs"], counts["improper_pairs"])
    if counts["without_qualities"]:
        logger.warning("Skipped %i reads without quality scores", counts["without_qualities"])
    for line in summary.log_lines():                                     # rescale.py:361-362
        logger.info("%s", line)
    logger.debug("Rescaling completed in %f seconds", time.time() - start)
    return 0


class _Ranks:
    """The ranks of a multi-GPU run (mapdamage/main.py:165-217 sharded by record): one process per GPU under
    torch.distributed; rank r of W takes the slabs (device decode) or the shard of every chunk (host decode) that are
    its own, nothing is exchanged until the tables are summed."""

    def __init__(self, options):
        import os
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.backend = options.dist_backend
        self.device = options.device
        if self.world > 1:
            import torch
            import torch.distributed as dist
            if not options.share_gpu:
                self.device = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(self.device)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if not dist.is_initialized():
                if self.backend == "nccl":
                    dist.init_process_group("nccl", device_id=torch.device("cuda", self.device))
                else:
                    dist.init_process_group("gloo")

    def finish(self, engine, error=None):
        """The tables of the whole run, on every rank.  ``error``: what this rank's part of the run died of, if it did —
        the ranks agree on it before any collective, the failing one re-raises it, the others raise RuntimeError."""
        if self.world == 1:
            if error is not None:
                raise error
            return engine.finish()
        from . import distributed
        if self.backend == "nccl":
            import torch
            dev = torch.device("cuda", self.device)
            distributed.agree_on_error(error, dev)
            return distributed.reduce_engine_tables(engine, dev)
        own = None
        if error is None:
            try:
                own = engine.finish()
            except (BadReadError, MdxError) as exc:
                error = exc
        distributed.agree_on_error(error)
        return distributed.reduce_tableset(own, engine.lgd_max)

    def close(self):
        if self.world > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.barrier()
                dist.destroy_process_group()


def launch_command(argv, gpus):
    """The command ``--gpus N`` re-executes itself under: one rank per GPU of this node."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    keep = [a for a in argv if a != "--print-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "mapdamage_amd"] + keep


def _tabulate_on_host(options, reader, ref, libraries, logger, ranks, carry=None):
    """The records decoded on the host (native BGZF/BAM decoder or SAM text), uploaded batch by batch.
    ``carry``: (engine, resume position, records counted so far) of a device decode that gave up part of the way: the
    same engine — its tables hold the slabs already counted — goes on with the rest of the file."""
    import contextlib
    with contextlib.ExitStack() as stack:
        if carry is None:
            engine = stack.enter_context(DamageEngine(libraries, options.length, options.around, options.minqual,
                                                      device=ranks.device))
            engine.set_reference(ref)
            n_reads, resume = 0, None
        else:
            engine, resume, n_reads = carry
            stack.enter_context(engine)
        warned_about_quals = False
        error = None
        # a BAM file arrives in chunks (bounded host memory; chunk k+1 is decoded while chunk k is tabulated)
        for batch in reader.iter_batches(resume=resume):
            if options.minqual and not warned_about_quals and batch.n:
                # main.py:185-192: the first iterated read without qualities (`not read.qual`: absent or
                # empty) triggers the warning, once
                lens = np.diff(batch.seq_off.astype(np.int64))
                first = np.minimum(batch.seq_off[:-1].astype(np.int64), max(0, batch.seq.shape[0] - 1))
                if batch.qual is None or bool(((lens == 0) | (batch.qual[first] == 0xFF)).any()):
                    logger.warning("Reads without PHRED scores found; cannot filter by --min-basequal")
                    warned_about_quals = True
            if options.minqual:
                # records none of whose qualities is below the threshold cannot be masked: the kernel skips their
                # quality windows; a chunk without a single maskable base goes through the unmasked kernel
                batch, nothing_to_mask = mark_unmaskable(batch, options.minqual)
                if nothing_to_mask:
                    batch = dataclasses.replace(batch, qual=None)
            # the slices of a chunk are enqueued one behind the other — the copy of slice k+1 runs under the kernel of
            # slice k — and the chunk is waited for once; a bad record comes back with its index among all records
            # iterated so far (the reference would name the read)
            # (several ranks: every rank decodes the chunk — the host decoder is one stream of records — and counts its
            # own contiguous shard of it)
            from .distributed import shard_bounds
            s_lo, s_hi = shard_bounds(batch.n, ranks.rank, ranks.world)
            try:
                for lo in range(s_lo, s_hi, options.batch_reads):
                    engine.tabulate(batch.slice(lo, min(s_hi, lo + options.batch_reads), copy=False), sync=False,
                                    record_base=n_reads + lo)
                engine.sync()
            except (BadReadError, MdxError) as exc:
                if ranks.world == 1:
                    raise
                error = exc
                break
            n_reads += batch.n
        tables = ranks.finish(engine, error)
    return tables


def _device_path_applies(options, world=1):
    """BAM files on disk; --downsample to a fraction too (the draws are made on the host from the flag column of every slab,
    reader.py:134-146) unless several ranks share the file (a rank steps over the slabs of the others without seeing their
    flags, and the stream of draws is the whole file's); a fixed number of reads is reservoir sampling over the whole file
    (reader.py:148-164): the host's."""
    from .sam import is_bam
    return str(options.filename) != "-" and is_bam(options.filename) and (
        options.downsample is None or (options.downsample < 1 and world == 1))


def _slab_bytes(options, world):
    """Compressed bytes per slab of the device decode path: a quarter of --chunk-mb (a slab of compressed bytes inflates to
    about four times its size) — 256 MiB at the default.  (Rounds 4-5 gave a single rank slabs of 1 GiB, for their fixed
    costs; since the slabs run as a pipeline of two — the device goes from one slab's inflate straight into the next one's —
    more, smaller slabs fill it better, and the pinned buffer of the host's share is a quarter of the size.)"""
    import os
    slab = max(1 << 20, int(options.chunk_mb * (1 << 20)) // 4) if options.chunk_mb else 256 << 20
    if os.environ.get("MDX_GBAM_SLAB_BYTES"):      # (tests: several slabs out of a small file whatever --chunk-mb says)
        slab = max(1 << 16, int(os.environ["MDX_GBAM_SLAB_BYTES"]))
    return slab


def _tabulate_on_device(options, reader, ref, libraries, logger, ranks, stages):
    """--gpu-decode: the file inflated, unpacked and counted on the GPU.  Returns (tables, None), or (None, carry) when
    the path does not apply or has given up — the caller decodes on the host, which also words the errors the way the
    reference does: the whole file (carry None), or, when the device path failed on a slab it had not begun to count,
    the rest of it with the same engine (``_tabulate_on_host``'s ``carry``)."""
    from .sam import GpuBamStream, GpuDecodeUnsupported
    if not _device_path_applies(options, ranks.world):
        logger.debug("the GPU decode path does not apply to this run; decoding on the host")
        return None, None
    import random
    downsample_rand = random.Random(options.downsample_seed)
    if options.merge_libraries:
        readgroups, lib_default = [], 0
    else:
        readgroups = [(rg, libraries.index(lib)) for rg, lib in reader._readgroups.items()]
        lib_default = None
    engine = DamageEngine(libraries, options.length, options.around, options.minqual, device=ranks.device)
    stages.mark("engine")
    carry = None
    try:
        engine.set_reference(ref)
        stages.mark("reference resident")
        warned_about_quals = False
        error = None
        slab = _slab_bytes(options, ranks.world)
        warm = getattr(options, "warm_thread", None)
        if warm is not None:
            warm.join()         # (the pinned buffer it leaves behind is the one the first slab takes)
        stages.mark("warm-up joined")
        with GpuBamStream(engine, options.filename, readgroups=readgroups, lib_default=lib_default,
                          chunk_bytes=slab, want_qual=options.minqual != 0, min_basequal=options.minqual) as stream:
            # (several ranks: rank r decodes the slabs r, r + W, ... and steps over the others)
            slab, n_reads = 0, 0
            try:
                while True:
                    mine = slab % ranks.world == ranks.rank
                    slab += 1
                    if not mine:
                        if not stream.skip():
                            break
                        continue
                    try:
                        view = stream.next_view()
                    except ValueError:
                        # the decode of a slab failed before any of its records was counted: everything in front of it
                        # is in the engine's tables, and the host decoder can go on from the slab's first record
                        if ranks.world == 1:
                            engine.sync()
                            where = stream.tell()
                            # (resuming needs the chunked host decoder, reader.iter_batches(resume=...): with --chunk-mb 0
                            # the host path reads the file in one piece, so the whole file is counted again)
                            # (... and --downsample: the host decoder starts its stream of draws at the file's first record)
                            if where is not None and slab > 1 and reader._chunks is not None and options.downsample is None:
                                carry = (engine, where, n_reads)
                        raise
                    if view is None:
                        break
                    if options.downsample is not None:
                        # reader.py:134-146: one draw per record the flag filter keeps, in file order, from the run's one
                        # generator; the records that leave get a bit the kernel's flag filter drops
                        flags = stream.view_flags(view)
                        kept = np.nonzero((flags & FLAG_FILTER) == 0)[0]
                        stay = draw_uniform(downsample_rand, len(kept)) < options.downsample
                        flags[kept[~stay]] |= 0x200
                        stream.set_view_flags(view, flags)
                        # (main.py:185-192 warns about a read without qualities that the loop MEETS — one that survived the draws:
                        # the decoder marks the records that have qualities, include/mdx.h MDX_FLAG_HAS_QUAL)
                        if options.minqual and not warned_about_quals and bool(((flags[kept[stay]] & 0x4000) == 0).any()):
                            logger.warning("Reads without PHRED scores found; cannot filter by --min-basequal")
                            warned_about_quals = True
                    elif options.minqual and not warned_about_quals and stream.missing_qualities():
                        logger.warning("Reads without PHRED scores found; cannot filter by --min-basequal")
                        warned_about_quals = True
                    engine.tabulate_view(view, record_base=n_reads)
                    n_reads += int(view.n_reads)
                engine.sync()
                stages.mark("decode and tabulate")
                if stream.fixups():
                    logger.debug("device decode: %d BGZF blocks rescanned from the record their predecessor's chain ended on", stream.fixups())
            except (BadReadError, ValueError, MdxError) as exc:
                if ranks.world == 1:
                    raise
                error = exc         # (the ranks agree on it in finish(): all of them take the host path then)
            tables = ranks.finish(engine, error)
        # (the stream is closed first: mdx_gbam_close hands its arena back to the context, which must still be alive)
        engine.close()
        return tables, None
    except GpuDecodeUnsupported as error:
        reason = "file layout the device path does not take (MDX_ERR_UNSUPPORTED): %s" % error
    except BadReadError as error:
        # a record the reference cannot process, or one without a usable read group: the host path names it
        reason = "a record the device path cannot count (MDX_ERR_BAD_READ, record %d)" % error.read_index
        carry = None
    except (ValueError, MdxError) as error:
        # a damaged file (the host decoder finds the same damage and words the error), or the device path out of
        # memory: either way the host path has the last word
        reason = "%s (libmdx code %s)" % (error, getattr(error, "code", "n/a"))
    except RuntimeError as error:
        # (several ranks: another rank's part of the file failed — every rank takes the host path, like that one)
        reason = str(error)
    # never silent: a regression of the device path must not show up as nothing but a slow run
    options.gpu_decode_fallbacks = getattr(options, "gpu_decode_fallbacks", 0) + 1
    if carry is None:
        engine.close()
        logger.warning("GPU decode path gave up: %s; decoding on the host (the whole file again)", reason)
    else:
        logger.warning("GPU decode path gave up: %s; decoding on the host (from compressed offset %d on: %d records are counted)",
                       reason, carry[1][0], carry[2])
    return None, carry


def main(argv):
    start_time = time.time()
    stages = _Stages()
    stages.mark("main")
    logging.basicConfig(format=_LOG_FORMAT, datefmt="%H:%M:%S")
    logger = logging.getLogger(__name__)
    try:
        options = parse_args(argv)
    except SystemExit as error:
        return int(error.code or 0) and 1
    import os
    if options.rescale_only:
        # the rescaling pass rewrites one BAM file in file order (rescale.py:285-365): one process, one GPU — under a launcher
        # that started several ranks the others leave at once instead of waiting in a process group for rank 0's whole pass
        if int(os.environ.get("RANK", "0")) != 0:
            return 0
        options.gpus = 1
        if os.environ.get("WORLD_SIZE", "1") != "1":
            for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
                os.environ.pop(key, None)
    if options.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one process per GPU: the run re-executes itself under torchrun (or is launched that way to begin with)
        import subprocess
        cmd = launch_command(list(argv), options.gpus)
        if options.print_launch:
            import json
            print(json.dumps(cmd))
            return 0
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        return subprocess.call(cmd, env=env)
    ranks = _Ranks(options)
    first = ranks.rank == 0
    if options.gpu_decode and not options.rescale_only and _device_path_applies(options, ranks.world) and not os.environ.get("MDX_NO_WARM"):
        # the device's context, the decode kernels and the pinned buffer of the host's share (about a fifth of a slab's
        # inflated bytes, which are four to five times its compressed ones), beside the header, index and FASTA reads
        import threading
        try:
            pinned = min(_slab_bytes(options, ranks.world), os.path.getsize(options.filename))
        except OSError:
            pinned = 0
        options.warm_thread = threading.Thread(target=_warm_up, args=(ranks.device, pinned), daemon=True)
        options.warm_thread.start()
    # (rank 0 keeps the log file and writes the tables; the other ranks speak up only when something is wrong)
    logging.getLogger().setLevel(options.log_level if first else "WARNING")
    handler = logging.FileHandler(options.folder / "Runtime_log.txt") if first else logging.NullHandler()
    handler.setFormatter(logging.Formatter(_LOG_FORMAT))
    handler.setLevel(options.log_level)
    logging.getLogger().addHandler(handler)
    try:
        logger.info("Started with the command: " + " ".join(sys.argv))
        if options.rescale_only:
            logger.info("Starting rescaling...")
            return rescale_qual(options) if first else 0
        if ranks.world > 1:
            logger.info("Rank 0 of %d: one process per GPU, records sharded by slab of the file, tables summed over %s",
                        ranks.world, "RCCL" if ranks.backend == "nccl" else "gloo (host)")
        # (the host is the node's: every rank takes its share of the threads cpu.max grants, include/mdx.h mdx_host_threads)
        from .engine import load_library
        from .sam import usable_cpus
        logger.debug("Host threads of this rank: %d inflating beside the device, %d for the host decoder (LOCAL_WORLD_SIZE %s)",
                     load_library().mdx_host_threads(), usable_cpus(), os.environ.get("LOCAL_WORLD_SIZE", "1"))
        reader = BAMReader(options.filename, merge_libraries=options.merge_libraries,
                           downsample_to=options.downsample, downsample_seed=options.downsample_seed,
                           chunk_bytes=int(options.chunk_mb * (1 << 20)))
        reflengths = reader.get_references()
        fai_lengths = read_fasta_index(str(options.ref) + ".fai")
        if not fai_lengths:
            return 1
        if not compare_sequence_dicts(fai_lengths, reflengths):
            return 1
        ref = reference_for_bam(options.ref, reader.handle.header.references)
        libraries = reader.get_libraries()
        stages.mark("headers and index")

        logger.info("Reading from '%s'", options.filename)
        if options.minqual != 0:
            logger.info("Filtering out bases with a Phred score < %d", options.minqual)
        logger.info("Writing results to '%s/'", options.folder)

        tables, carry = _tabulate_on_device(options, reader, ref, libraries, logger, ranks, stages) if options.gpu_decode else (None, None)
        if tables is None:
            tables = _tabulate_on_host(options, reader, ref, libraries, logger, ranks, carry)
        fallbacks = getattr(options, "gpu_decode_fallbacks", 0)
        if options.gpu_decode:
            logger.log(logging.WARNING if fallbacks else logging.DEBUG, "Decode path: %s; fallbacks from the device path: %d",
                       "host decoder" if (fallbacks or not _device_path_applies(options, ranks.world)) else "device", fallbacks)
        logger.debug("Done. %d filtered alignments processed", tables.n_kept)
        logger.debug("BAM read in %f seconds", time.time() - start_time)

        stages.mark("tables")
        if not first:
            return 0
        tables.write(options.folder)
        if options.freq_files:
            (options.folder / "5pCtoT_freq.txt").write_text(tables.damage_frequency_text("5p", options.readplot))
            (options.folder / "3pGtoA_freq.txt").write_text(tables.damage_frequency_text("3p", options.readplot))
        check_table_and_warn_if_dmg_freq_is_low(options.folder)
        logger.info("Successful run")
        logger.debug("Run completed in %f seconds", time.time() - start_time)
        stages.mark("files written")
        return 0
    except BadReadError as error:
        # the reference dies with pysam's ValueError here (align.py:33)
        logger.error("%s", error)
        raise
    except BAMError as error:
        logger.error("%s", error)
        raise
    finally:
        logging.getLogger().removeHandler(handler)
        handler.close()
        ranks.close()
        stages.mark("end")
        stages.write()


def entry_point():
    """The command's exit: the tables are on disk and the log is closed when ``main`` returns, and what is left — the
    interpreter's and the HIP runtime's teardown, unpinning and unmapping a few gigabytes — is a fifth of a second the
    operating system does faster for a process that simply leaves (MDX_NO_FAST_EXIT=1: the ordinary way out)."""
    import os
    rc = main(sys.argv[1:])
    if os.environ.get("MDX_NO_FAST_EXIT"):
        return rc
    logging.shutdown()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(int(rc or 0))


if __name__ == "__main__":
    sys.exit(entry_point())
