#!/usr/bin/env python3
"""bench.py — reads/s of the damage-tabulation hot path on N MI355X (one process per GPU).

A "step" is one pass of the hot path over one resident batch of synthetic reads: survey
config 2 (BASELINE.json configs[1]) — 5 M single-end 100 bp `100M` reads with C>T / G>A
damage, --length 70 --around 10, one library, genome of 10 Mb resident in HBM.  Weak
scaling: every rank tabulates its own 5 M-read batch; the count tables are summed with one
RCCL all-reduce at the end of the timed region.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch
(240 B/read x reads) / average tabulation-kernel duration measured with HIP
events on the launch stream.  `cpu_baseline` = the C oracle (oracle/mdx_oracle.c) on the same
batch on the host's threads (`cores`; the one-thread rate is quoted in `sample`), which also
serves as the bit-exactness check of the GPU tables.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_READ = 240  # 100 (SEQ) + 100 + 2*10 (reference slice + flanks) + 4 (CIGAR) + 16
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=5_000_000, help="reads per GPU per step")
    ap.add_argument("--cpu-reads", type=int, default=5_000_000, help="bounded CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even for one rank: exercises the all-reduce path")
    ap.add_argument("--sorted", action="store_true", help="coordinate-sort the batch (like a sorted BAM)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4),
                    help="survey workload: 2 = headline (SE 100M reads); 3 = paired, clips + indels; 4 = 35-150 bp")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if (world > 1 or args.force_dist) else 0)

    from mapdamage_amd import build, synth
    if rank == 0:
        build.build_lib()
    if world > 1:
        dist.barrier()
    from mapdamage_amd.engine import DamageEngine
    from mapdamage_amd.distributed import allreduce_words

    L, A = 70, 10
    ref = synth.make_genome()
    if args.config == 2:
        batch = synth.config2_batch(ref, args.reads, seed=2 + rank)
    elif args.config == 3:
        batch = synth.config3_batch(ref, args.reads, seed=3 + rank)
    else:
        batch = synth.config4_batch(ref, args.reads, seed=4 + rank)
    if args.sorted:
        import numpy as _np
        batch = synth._permute_fixed(batch, _np.lexsort((batch.pos, batch.tid)))
    libs = [("synthetic", "lib1")]
    eng = DamageEngine(libs, L, A, 0, lgd_max=4096, device=dev.index)
    eng.set_reference(ref)
    dbatch = eng.upload(batch)
    tables = torch.zeros(eng.table_words(), dtype=torch.int64, device=dev)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    for _ in range(args.warmup):
        eng.tabulate(dbatch)
    barrier()
    eng.reset()
    eng.timing(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.tabulate(dbatch)
    eng.finish_device(tables.data_ptr())
    eng.sync()
    allreduce_words(tables)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_launch, kernel_ms = eng.timing_read()

    # HBM traffic per launch from the committed PMC summary of this same command (tools/prof.sh)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            tj = json.load(fh)
        if tj.get("reads_per_launch") == args.reads:
            traffic = (tj["FETCH_SIZE_KB"] + tj["WRITE_SIZE_KB"]) * 1024.0
    except (OSError, ValueError, KeyError):
        pass

    total_reads = args.reads * args.steps * world
    value = total_reads / dt
    per_launch_ms = kernel_ms / max(1, n_launch)
    algo_bytes = ALGO_BYTES_PER_READ * args.reads
    if args.config != 2:
        # qlen + (reflen + 2A) + 4 n_cigar + 16 per record, reflen ~ aligned query here
        algo_bytes = float(2 * batch.seq.shape[0] + 4 * batch.cigar.shape[0] + (2 * A + 16) * batch.n)
    achieved = algo_bytes / (per_launch_ms * 1e-3) / 1e9

    out = {
        "metric": "reads/sec (misincorporation+comp tables)",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": ("config2: %d SE 100bp 100M reads/GPU, C>T/G>A damage, --length 70 "
                                "--around 10, 1 library, 10 Mb genome resident" % args.reads) if args.config == 2
                   else "config%d (survey §8d), %d reads/GPU, --length 70 --around 10" % (args.config, args.reads),
                   "reads_per_gpu": args.reads, "length": L, "around": A,
                   "parallelism": "shard-by-read x%d + RCCL all-reduce of tables" % world,
                   "table_mode": eng.table_mode},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "tabulate_kernel", "kernel_ms": per_launch_ms,
                     "algorithmic_bytes_per_read": ALGO_BYTES_PER_READ},
    }

    if rank == 0:
        words = tables.cpu().numpy().view(np.uint64)
        got = eng.unpack_tables(words)
        if not args.no_cpu and world == 1:
            # CPU baseline = the oracle (a port: the reference's Python path cannot travel), on rank 0's own
            # batch (bounded sample), N = 1 only: all host threads (`cores`), and one thread on a tenth of it
            from oracle import oracle
            n_cpu = min(args.cpu_reads, batch.n)
            sample = batch if n_cpu == batch.n else batch.slice(0, n_cpu)
            t1 = time.perf_counter()
            want, n_thr = oracle.tabulate_parallel(ref, sample, 1, L, A, 0, lgd_max=4096)
            cpu_dt = time.perf_counter() - t1
            n_one = max(1, n_cpu // 10)
            t1 = time.perf_counter()
            oracle.tabulate(ref, batch.slice(0, n_one), 1, L, A, 0, lgd_max=4096)
            one_dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": n_cpu / cpu_dt, "unit": "reads/s", "cores": n_thr,
                                   "kind": "port",
                                   "sample": "%d reads of the same config-%d batch, C oracle on %d threads "
                                             "(1 thread, %d reads: %.0f reads/s)"
                                             % (n_cpu, args.config, n_thr, n_one, n_one / one_dt)}
            if world == 1 and n_cpu == batch.n:
                ok = (np.array_equal(got.mis, want["mis"] * np.uint64(args.steps))
                      and np.array_equal(got.comp, want["comp"] * np.uint64(args.steps))
                      and np.array_equal(got.lgd, want["lgd"] * np.uint64(args.steps)))
                out["parity"] = "bit-exact vs oracle" if ok else "MISMATCH"
                if not ok:
                    print(json.dumps(out))
                    raise SystemExit("GPU tables differ from the oracle")
        assert got.n_kept == total_reads, (got.n_kept, total_reads)
    dbatch.free()
    eng.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line, last thing on stdout: RCCL's version banner (NCCL_DEBUG=VERSION) sits in the
        # C stdio buffer until exit, so flush the C streams first
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        sys.stderr.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
