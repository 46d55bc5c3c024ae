"""File -> tables through the device decode path for several library builds and slab sizes: a config-3 BAM written once
(forked workers, before the GPU is touched), then best of `reps` runs per (build, slab), tables checked against the
first build's.  usage: python tools/decode_ab.py [--reads N] [--slabs "32,64,128,256"] tag1 tag2 ...  ("cur" = in-tree)"""
import argparse
import os
import pathlib
import sys
import tempfile
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

from mapdamage_amd import engine, sam, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tags", nargs="+")
    ap.add_argument("--reads", type=int, default=8_000_000)
    ap.add_argument("--slabs", default="32,64,128,256")
    ap.add_argument("--reps", type=int, default=4)
    args = ap.parse_args()
    ref = synth.make_genome()
    batch = synth.parallel_batch("config3_batch", ref, args.reads, seed=3003, workers=64)
    tmp = tempfile.mkdtemp(prefix="mdx_dab_")
    path = os.path.join(tmp, "c3.bam")
    sam.write_bam(path, batch, ref.names, ref.lengths, [{"ID": "rg1", "SM": "synthetic", "LB": "lib1"}],
                  rg_of_record=["rg1"] * batch.n, workers=64)
    with open(path, "rb") as fh:
        while fh.read(1 << 26):
            pass
    first = None
    for tag in args.tags:
        lib = ROOT / "mapdamage_amd" / "libmdx.so" if tag == "cur" else ROOT / "tools" / "bin" / ("libmdx_%s.so" % tag)
        engine._lib = engine.load_library(str(lib))
        with engine.DamageEngine([("synthetic", "lib1")], 70, 10, 0) as eng:
            eng.set_reference(ref)
            row = []
            for mb in [int(x) for x in args.slabs.split(",")]:
                best = 1e9
                for rep in range(args.reps + 1):
                    eng.reset()
                    eng.sync()
                    t0 = time.perf_counter()
                    with sam.GpuBamStream(eng, path, readgroups=[("rg1", 0)], chunk_bytes=mb << 20) as g:
                        while True:
                            v = g.next_view()
                            if v is None:
                                break
                            eng.tabulate_view(v)
                        got = eng.finish()
                    dt = time.perf_counter() - t0
                    if rep:
                        best = min(best, dt)
                if first is None:
                    first = got
                assert np.array_equal(got.mis, first.mis) and np.array_equal(got.comp, first.comp) and got.n_kept == first.n_kept
                row.append("%d MiB: %.1f ms = %.1f M reads/s" % (mb, best * 1e3, args.reads / best / 1e6))
        print(tag, " | ".join(row), flush=True)
    os.remove(path)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
