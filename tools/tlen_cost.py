"""Kernel time by insert size: fragment lengths below 256 are counted in the LDS, longer ones with global atomics
on the dense histogram (2 M paired 100 bp records).  Run on the GPU box."""
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
    ref = synth.make_genome()
    b = synth.make_reads(ref, n, 3, read_len=100, paired=True, contigs=[0, 1])
    rng = np.random.default_rng(1)
    for mean in (180, 350, 600, 3000):
        t = np.maximum(100, rng.normal(mean, mean / 6, n)).astype(np.int32)
        b.tlen = np.where(b.tlen < 0, -t, t).astype(np.int32)
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
            print(json.dumps({"mean_insert": mean, "kernel_ms": ms / 5, "Greads_per_s": n / (ms / 5 * 1e-3) / 1e9}), flush=True)


if __name__ == "__main__":
    main()
