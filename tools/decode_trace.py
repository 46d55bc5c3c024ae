"""Where the host thread is while a BAM file goes through the device decode path: a config-3 BAM written once (forked workers,
before the GPU is touched), a warm-up pass, then passes with MDX_BAM_TRACE=1 (mdx_gbam_next prints when it reached each point
of the call; nothing is synchronised that the call would not wait for anyway) and the Python-side times of every call.
usage: python tools/decode_trace.py [--reads N] [--slab-mb M] [--reps R] [--cli]"""
import argparse
import json
import os
import pathlib
import subprocess
import sys
import tempfile
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import engine, fasta, sam, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=24_000_000)
    ap.add_argument("--slab-mb", type=int, default=256)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--cli", action="store_true", help="also: the command line as a cold process, with its stage stamps")
    args = ap.parse_args()
    ref = synth.make_genome()
    workers = min(16, sam.usable_cpus())
    batch = synth.parallel_batch("config3_batch", ref, args.reads, seed=3003, workers=workers)
    tmp = tempfile.mkdtemp(prefix="mdx_trace_")
    path = os.path.join(tmp, "c3.bam")
    sam.write_bam(path, batch, ref.names, ref.lengths, [{"ID": "rg1", "SM": "synthetic", "LB": "lib1"}], rg_of_record="rg1", workers=workers)
    fasta.write_fasta(os.path.join(tmp, "ref.fa"), ref)
    with open(path, "rb") as fh:
        while fh.read(1 << 26):
            pass
    print("file: %d records, %.1f MB" % (batch.n, os.path.getsize(path) / 1e6), file=sys.stderr)
    with engine.DamageEngine([("synthetic", "lib1")], 70, 10, 0) as eng:
        eng.set_reference(ref)
        # rep 0: warm-up; then `reps` traced passes, `reps` plain ones, and `reps` with one slab at a time (MDX_GBAM_NO_LOOKAHEAD)
        for rep in range(3 * args.reps + 1):
            os.environ.pop("MDX_BAM_TRACE", None)
            os.environ.pop("MDX_GBAM_NO_LOOKAHEAD", None)
            if 1 <= rep <= args.reps:
                os.environ["MDX_BAM_TRACE"] = "1"
            if rep > 2 * args.reps:
                os.environ["MDX_GBAM_NO_LOOKAHEAD"] = "1"
            eng.reset()
            eng.sync()
            t0 = time.perf_counter()
            calls = []
            with sam.GpuBamStream(eng, path, readgroups=[("rg1", 0)], chunk_bytes=args.slab_mb << 20) as g:
                t_open = time.perf_counter()
                while True:
                    t1 = time.perf_counter()
                    v = g.next_view()
                    t2 = time.perf_counter()
                    if v is None:
                        break
                    eng.tabulate_view(v)
                    calls.append((round(1e3 * (t2 - t1), 2), round(1e3 * (time.perf_counter() - t2), 2)))
                t3 = time.perf_counter()
                got = eng.finish()
                t4 = time.perf_counter()
            t5 = time.perf_counter()
            print("rep %d%s: open %.2f ms, (next_view, tabulate_view) %s, last next %.2f, finish %.2f, close %.2f; open..tables %.1f ms = %.1f M reads/s"
                  % (rep, " (no lookahead)" if rep > 2 * args.reps else "", 1e3 * (t_open - t0), calls, 1e3 * (t3 - t1), 1e3 * (t4 - t3), 1e3 * (t5 - t4), 1e3 * (t4 - t0), args.reads / (t4 - t0) / 1e6),
                  file=sys.stderr, flush=True)
            assert got.n_kept == args.reads
    os.environ.pop("MDX_BAM_TRACE", None)
    os.environ.pop("MDX_GBAM_NO_LOOKAHEAD", None)
    if args.cli:
        for k in range(6):
            out_dir = os.path.join(tmp, "out%d" % k)
            env = dict(os.environ, MDX_STAGE_LOG=os.path.join(tmp, "stages.json"))
            if k >= 4:
                env["MDX_INIT_TRACE"] = "1"
            if k in (2, 3):
                env["MDX_GBAM_SLAB_BYTES"] = str(1 << 30)
            t_spawn, t0 = time.time(), time.perf_counter()
            subprocess.run([sys.executable, "-m", "mapdamage_amd", "-i", path, "-r", os.path.join(tmp, "ref.fa"), "-d", out_dir, "--no-stats"],
                           cwd=str(ROOT), env=env, check=True, stdout=subprocess.DEVNULL)
            wall = time.perf_counter() - t0
            marks = json.load(open(os.path.join(tmp, "stages.json")))["stages"]
            stamps = [("spawn", t_spawn)] + [tuple(m) for m in marks] + [("exit", t_spawn + wall)]
            print("cli run %d%s: wall %.3f s; " % (k, " (1 GiB slabs)" if k in (2, 3) else "", wall) + ", ".join("%s %.0f ms" % (stamps[i][0], 1e3 * (stamps[i][1] - stamps[i - 1][1])) for i in range(1, len(stamps))),
                  file=sys.stderr, flush=True)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
