"""Kernel time of the tabulation pass by number of libraries (config-3 records dealt to the libraries at random, as the
read groups of a BAM file are; reader.py:47-50, statistics.py:12-20): the packed kernel over a resident 4-bit batch —
ONE launch, a library per pool of blocks over the records bucketed by library (mdx_batch::libsort, built at upload) —, the same
with the sort inside every launch (a batch that does not bring it), and the ASCII kernel (one launch per group of
libraries that fits the LDS, each over all records).  MDX_NO_ML=1 in the environment: the packed kernel as it was
before round 5, one launch per library over all records.  Run on the GPU box:
    python tools/nlib_cost.py [--min-basequal Q] [records] [libraries ...]
--min-basequal Q: the same with qualities (5 % of the bases below Phred 20) through a context at that threshold — the packed
masked kernels (the mask in the resident column, MDX_SEQ_4BITQ); the ASCII kernel and the sort inside the launch are left out."""
import ctypes
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.engine import DamageEngine  # noqa: E402


def timed(eng, what, reps=5):
    what()
    eng.sync()
    eng.timing(True)
    for _ in range(reps):
        what()
    eng.sync()
    n_launch, ms = eng.timing_read()
    eng.timing(False)
    return n_launch // reps, ms / reps


def main():
    argv = list(sys.argv[1:])
    minqual = 0
    if "--min-basequal" in argv:
        at = argv.index("--min-basequal")
        minqual = int(argv[at + 1])
        del argv[at:at + 2]
    n = int(argv[0]) if argv else 16_000_000
    counts = [int(x) for x in argv[1:]] or [1, 2, 4, 8]
    ref = synth.make_genome()
    base = None
    for nlib in counts:
        libs = [("s", "l%d" % i) for i in range(nlib)]
        b = synth.make_reads(ref, n, 3, read_len=100, nlib=nlib, contigs=[0, 1], paired=True, frac_softclip=0.10,
                             frac_ins=0.04, frac_del=0.04, frac_skip=0.002, frac_hardclip=0.001, with_qual=minqual > 0)
        row = {"records": n, "libraries": nlib, "min_basequal": minqual}
        if minqual:
            import numpy as np
            rng = np.random.default_rng(2020)
            low = rng.random(b.qual.shape[0]) < 0.05
            b.qual = np.where(low, rng.integers(2, 20, b.qual.shape[0]), rng.integers(30, 42, b.qual.shape[0])).astype(np.uint8)
        with DamageEngine(libs, 70, 10, minqual, lgd_max=4096) as eng:
            eng.set_reference(ref)
            db = eng.upload(b, packed=True)
            row["packed_launches_per_pass"], row["packed_ms"] = timed(eng, lambda: eng.tabulate(db))
            if minqual:
                db.free()
                if nlib == 1:
                    base = row["packed_ms"]
                if base:
                    row["packed_x_one_library"] = round(row["packed_ms"] / base, 3)
                print(json.dumps(row), flush=True)
                continue
            if nlib > 1:
                view = type(db.dev)()
                ctypes.memmove(ctypes.byref(view), ctypes.byref(db.dev), ctypes.sizeof(view))
                view.libsort = None
                _, row["packed_sort_in_launch_ms"] = timed(eng, lambda: eng.tabulate_view(view))
            db.free()
            da = eng.upload(b, packed=False)
            row["ascii_launches_per_pass"], row["ascii_ms"] = timed(eng, lambda: eng.tabulate(da))
            da.free()
        if nlib == 2:
            # (the several-library kernel itself: two libraries, every record in the first)
            b.lib[:] = 0
            with DamageEngine(libs, 70, 10, 0, lgd_max=4096) as eng:
                eng.set_reference(ref)
                db = eng.upload(b, packed=True)
                _, row["packed_all_in_library_0_ms"] = timed(eng, lambda: eng.tabulate(db))
                db.free()
        if nlib == 1:
            base = row["packed_ms"]
        if base:
            row["packed_x_one_library"] = round(row["packed_ms"] / base, 3)
        row["packed_Greads_per_s"] = round(n / (row["packed_ms"] * 1e-3) / 1e9, 3)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
