"""Kernel time of the tabulation pass by number of libraries (2 M config-2-style records): one library keeps its
tables in the LDS; more than one does not fit next to the staging areas.  Run on the GPU box."""
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.engine import DamageEngine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    ref = synth.make_genome()
    for nlib in (1, 2, 3, 8):
        libs = [("s", "l%d" % i) for i in range(nlib)]
        b = synth.make_reads(ref, n, 2, read_len=100, nlib=nlib, contigs=[0, 1])
        with DamageEngine(libs, 70, 10, 0, lgd_max=4096) as eng:
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
            print(json.dumps({"libraries": nlib, "table_mode": eng.table_mode, "launches_per_pass": n_launch // 5,
                              "kernel_ms": ms / 5, "Greads_per_s": n / (ms / 5 * 1e-3) / 1e9}), flush=True)


if __name__ == "__main__":
    main()
