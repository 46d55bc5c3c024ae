"""Stage times of the native BAM decoder (MDX_BAM_TIMING=1) on a synthetic config-3 BAM: whole-file decode, then
the chunked decoder alone, then chunked + flag filter + library column (no GPU work).  Run on the GPU box (host
cores only): python tools/decode_timing.py [reads]"""
import os
import pathlib
import sys
import tempfile
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["MDX_BAM_TIMING"] = "1"

from mapdamage_amd import sam, synth  # noqa: E402
from mapdamage_amd.engine import load_library  # noqa: E402
from mapdamage_amd.reader import BAMReader  # noqa: E402

load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
ref = synth.make_genome()
batch = synth.config3_batch(ref, n, seed=3)
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "c3.bam")
    sam.write_bam(path, batch, ref.names, ref.lengths, [{"ID": "rg1", "SM": "synthetic", "LB": "lib1"}],
                  rg_of_record=["rg1"] * n)
    for threads in (int(os.environ.get("MDX_THREADS", "64")), 32):
        t = time.perf_counter()
        al = sam.read_bam_native(path, threads=threads)
        print("whole file, threads %d: total %.1f ms (%d records)" % (threads, 1e3 * (time.perf_counter() - t), al.batch.n), file=sys.stderr)
        del al
        time.sleep(0.5)
    t = time.perf_counter()
    with sam.BamStream(path, chunk_bytes=256 << 20) as stream:
        while True:
            t1 = time.perf_counter()
            chunk = stream.next_chunk()
            print("next_chunk %.1f ms" % (1e3 * (time.perf_counter() - t1)), file=sys.stderr)
            if chunk is None:
                break
    print("chunked decode alone: total %.1f ms" % (1e3 * (time.perf_counter() - t)), file=sys.stderr)
    time.sleep(0.5)
    os.environ.pop("MDX_BAM_TIMING")
    t = time.perf_counter()
    k = sum(b.n for b in BAMReader(path, chunk_bytes=256 << 20).iter_batches())
    print("BAMReader.iter_batches (decode thread + filter + library column): total %.1f ms (%d records)"
          % (1e3 * (time.perf_counter() - t), k), file=sys.stderr)
