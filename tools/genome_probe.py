"""Is the tabulation kernel bound by the reference gathers?  Kernel-only time of plain 100 bp records over genomes
of 1 Mb (fits one XCD's L2), 10 Mb (the survey's: Infinity Cache) and 400 Mb (beyond the Infinity Cache), unsorted and
coordinate-sorted.  Run on the GPU box: python tools/genome_probe.py [reads]"""
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.engine import DamageEngine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    cases = []
    for mb in (1, 10, 400):
        ref = synth.make_genome(sizes=(("chr1", mb * 800_000), ("chr2", mb * 200_000), ("chrS", 500)))
        b = synth.parallel_batch(dict(read_len=100, paired=True, contigs=[0, 1]), ref, n, 3, workers=64)
        cases.append((mb, ref, b, synth._permute_fixed(b, np.lexsort((b.pos, b.tid)))))
    for mb, ref, b, bs in cases:
        with DamageEngine([("s", "l")], 70, 10, 0, lgd_max=4096) as eng:
            eng.set_reference(ref)
            for name, batch in (("unsorted", b), ("sorted", bs)):
                db = eng.upload(batch)
                eng.tabulate(db)
                eng.sync()
                eng.timing(True)
                for _ in range(20):
                    eng.tabulate(db)
                eng.sync()
                nl, ms = eng.timing_read()
                eng.timing(False)
                db.free()
                print(json.dumps({"genome_mb": mb, "order": name, "reads": n, "kernel_ms": ms / nl,
                                  "ms_per_2M": ms / nl * 2e6 / n}), flush=True)


if __name__ == "__main__":
    main()
