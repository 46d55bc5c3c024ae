"""BASELINE configs[2] at full size: 50 M config-3 records (paired flags, soft clips, indels, N ops) on one
MI355X as two resident batches of 25 M (the seq_off column is 32-bit: 4 GiB of bases per batch),
accumulated in one context and compared bit for bit with the C oracle (all host threads).
Run on the GPU box: python tools/fullsize_check.py [reads_per_batch] [batches] [config 3|4]
(config 4 = one GPU's 50 M-record share of BASELINE configs[3], lengths 35-150)."""
import json
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.engine import DamageEngine  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    config = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    make = synth.config3_batch if config == 3 else synth.config4_batch
    L, A = 70, 10
    ref = synth.make_genome()
    libs = [("synthetic", "lib1")]
    want = None
    t_gen = t_cpu = 0.0
    with DamageEngine(libs, L, A, 0, lgd_max=4096) as eng:
        eng.set_reference(ref)
        eng.timing(True)
        for b in range(nb):
            t0 = time.perf_counter()
            batch = make(ref, per, seed=config + 100 * b)
            t_gen += time.perf_counter() - t0
            dev = eng.upload(batch)
            eng.tabulate(dev)
            eng.sync()
            dev.free()
            t0 = time.perf_counter()
            part, threads = oracle.tabulate_parallel(ref, batch, 1, L, A, 0, lgd_max=4096)
            t_cpu += time.perf_counter() - t0
            if want is None:
                want = part
            else:
                for k in ("mis", "comp", "lgd"):
                    want[k] += part[k]
                want["n_kept"] += part["n_kept"]
            del batch
        n_launch, kernel_ms = eng.timing_read()
        got = eng.finish()
    ok = (np.array_equal(got.mis, want["mis"]) and np.array_equal(got.comp, want["comp"])
          and np.array_equal(got.lgd, want["lgd"]) and got.n_kept == want["n_kept"])
    print(json.dumps({"workload": "config%d (survey 8d), %d batches x %d records" % (config, nb, per),
                      "records": per * nb, "kept": int(got.n_kept), "parity": "bit-exact vs oracle" if ok else "MISMATCH",
                      "kernel_ms_total": kernel_ms, "launches": n_launch,
                      "gpu_reads_per_s_kernel": per * nb / (kernel_ms * 1e-3),
                      "oracle_threads": threads, "oracle_reads_per_s": per * nb / t_cpu,
                      "mis_sum": int(got.mis.sum()), "comp_sum": int(got.comp.sum())}))
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
