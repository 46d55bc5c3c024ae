"""End-to-end rate from a BAM file on disk to the three tables (N1 row): synthetic config-3 records with one read
group are written as a BAM, then timed stage by stage — native multi-threaded BGZF/BAM decode (mdx_bam_*), flag
filter + library column (reader.py), tabulation from host buffers (mdx_tabulate_host: H2D + kernel), finish; then the
same file through the chunked host decoder (the command line's --host-decode path) and through the GPU decode path
(mdx_gbam_*: compressed bytes to HBM, inflate + CRC32 + unpack + tabulation on the device; the command line's default).
The tables of every variant are checked against the C oracle.  Run on the GPU box: python tools/e2e_bench.py [reads]"""
import json
import os
import pathlib
import sys
import tempfile
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

from mapdamage_amd import sam, synth  # noqa: E402
from mapdamage_amd.engine import DamageEngine  # noqa: E402
from mapdamage_amd.reader import BAMReader  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    ref = synth.make_genome()
    batch = synth.config3_batch(ref, n, seed=3)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "config3.bam")
        sam.write_bam(path, batch, ref.names, ref.lengths, [{"ID": "rg1", "SM": "synthetic", "LB": "lib1"}],
                      rg_of_record=["rg1"] * n)
        size = os.path.getsize(path)
        t0 = time.perf_counter()
        reader = BAMReader(path)
        t1 = time.perf_counter()
        idx = reader.kept_indices()
        b = reader.handle.batch
        if len(idx) != b.n:
            b = b.take(idx)
        b.lib = reader.library_column(idx)
        libs = reader.get_libraries()
        t2 = time.perf_counter()
        with DamageEngine(libs, 70, 10, 0) as eng:
            eng.set_reference(ref)
            eng.tabulate(b.slice(0, min(b.n, 1000)))   # warm-up (module load, staging allocation)
            eng.sync()
            eng.reset()
            t3 = time.perf_counter()
            step = 4_000_000
            for lo in range(0, b.n, step):
                eng.tabulate(b.slice(lo, lo + step, copy=False))
            got = eng.finish()
            t4 = time.perf_counter()
            # the same file in chunks: chunk k+1 is decoded on a helper thread while chunk k is filtered and
            # tabulated (the path of the command line, mapdamage_amd/main.py); peak host memory = two chunks
            streamed = {}
            for chunk_mb in (64, 256):
                eng.reset()
                t5 = time.perf_counter()
                chunked = BAMReader(path, chunk_bytes=chunk_mb << 20)
                n_chunks = 0
                for part in chunked.iter_batches():
                    n_chunks += 1
                    for lo in range(0, part.n, step):
                        eng.tabulate(part.slice(lo, lo + step, copy=False))
                got_s = eng.finish()
                t6 = time.perf_counter()
                streamed[chunk_mb] = (t6 - t5, n_chunks, got_s)
            # the same file decoded on the GPU (mdx_gbam_*): compressed bytes to HBM, inflate + unpack + tabulation there
            gpu = {}
            for chunk_mb in (64, 256):
                eng.reset()
                eng.sync()
                t7 = time.perf_counter()
                n_chunks = 0
                with sam.GpuBamStream(eng, path, readgroups=[("rg1", 0)], chunk_bytes=chunk_mb << 20) as g:
                    while True:
                        view = g.next_view()
                        if view is None:
                            break
                        n_chunks += 1
                        eng.tabulate_view(view)
                    got_g = eng.finish()
                t8 = time.perf_counter()
                gpu[chunk_mb] = (t8 - t7, n_chunks, got_g)
    want = oracle.tabulate(ref, batch, 1, 70, 10, 0, 65536)
    ok = (np.array_equal(got.mis, want["mis"]) and np.array_equal(got.comp, want["comp"]) and got.n_kept == want["n_kept"])
    for chunk_mb, (_, _, got_s) in streamed.items():
        ok = ok and np.array_equal(got_s.mis, want["mis"]) and np.array_equal(got_s.comp, want["comp"]) \
            and np.array_equal(got_s.lgd, got.lgd) and got_s.n_kept == want["n_kept"]
    for chunk_mb, (_, _, got_g) in gpu.items():
        ok = ok and np.array_equal(got_g.mis, want["mis"]) and np.array_equal(got_g.comp, want["comp"]) \
            and np.array_equal(got_g.lgd, got.lgd) and got_g.n_kept == want["n_kept"]
    print(json.dumps({
        "workload": "config 3, %d records, BAM %.1f MB (BGZF level of sam.write_bam)" % (n, size / 1e6),
        "decode_s": t1 - t0, "decode_reads_per_s": n / (t1 - t0), "host_threads": min(64, os.cpu_count() or 1),
        "filter_library_s": t2 - t1,
        "tabulate_host_s": t4 - t3, "tabulate_host_reads_per_s": b.n / (t4 - t3),
        "end_to_end_reads_per_s": n / ((t2 - t0) + (t4 - t3)),
        "gpu_decode": {"%d MiB compressed per slab" % mb: {"slabs": k, "s": dt, "end_to_end_reads_per_s": n / dt}
                       for mb, (dt, k, _) in gpu.items()},
        "streamed": {"%d MiB chunks" % mb: {"chunks": k, "s": dt, "end_to_end_reads_per_s": n / dt}
                     for mb, (dt, k, _) in streamed.items()},
        "parity": "bit-exact vs oracle" if ok else "MISMATCH"}))
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
