"""Kernel time by genome size: the benchmark genome (10 Mb) stays in the L2 / Infinity Cache; the reference windows
of a large genome come from HBM — at random for an unsorted batch, nearly sequentially for a coordinate-sorted one.
2 M config-2 records.  Run on the GPU box."""
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.engine import DamageEngine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    for mb in (10, 100, 1000):
        ref = synth.make_genome(sizes=(("chr1", mb * 800_000), ("chr2", mb * 200_000), ("chrS", 500)))
        b = synth.make_reads(ref, n, 2, read_len=100, contigs=[0, 1])
        for order in ("unsorted", "sorted"):
            if order == "sorted":
                b = synth._permute_fixed(b, np.lexsort((b.pos, b.tid)))
            with DamageEngine([("s", "l")], 70, 10, 0, lgd_max=4096) as eng:
                eng.set_reference(ref)
                db = eng.upload(b)
                eng.tabulate(db)
                eng.sync()
                eng.timing(True)
                for _ in range(5):
                    eng.tabulate(db)
                eng.sync()
                n_launch, ms = eng.timing_read()
                db.free()
                print(json.dumps({"genome_Mb": mb, "batch": order, "kernel_ms": ms / 5,
                                  "Greads_per_s": n / (ms / 5 * 1e-3) / 1e9}), flush=True)


if __name__ == "__main__":
    main()
