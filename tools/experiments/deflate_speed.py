"""Speed of the device BGZF writer (mdx_bgzf_deflate) on a stream of config-5 records, against zlib level 6 on the host's threads."""
import gzip, os, sys, time, zlib
sys.path.insert(0, os.getcwd())
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from mapdamage_amd import sam, synth
from mapdamage_amd.engine import DamageEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
ref = synth.make_genome()
b = synth.parallel_batch(dict(read_len=100, paired=True, frac_softclip=0.10, frac_ins=0.04, frac_del=0.04, frac_skip=0.002, frac_hardclip=0.0,
                              contigs=[0, 1], with_qual=True), ref, n, seed=5005, workers=16)
path = "/tmp/ds.bam"
sam.write_bam(path, b, ref.names, ref.lengths, [{"ID": "rg1", "SM": "s", "LB": "l"}], rg_of_record="rg1", workers=16)
stream = gzip.decompress(open(path, "rb").read())
print("stream %d bytes, zlib-6 file %d bytes" % (len(stream), os.path.getsize(path)), flush=True)
with DamageEngine([("s", "l")]) as eng:
    for rep in range(3):
        t0 = time.perf_counter()
        out = eng.bgzf_deflate(stream)
        dt = time.perf_counter() - t0
        print("device: %.3f s, %.2f GB/s in, %d bytes out (%.3f of zlib-6)" % (dt, len(stream) / dt / 1e9, len(out), len(out) / os.path.getsize(path)), flush=True)
    assert gzip.decompress(bytes(out)) == stream
blocks = [stream[i:i + 0xFF00] for i in range(0, len(stream), 0xFF00)]
t0 = time.perf_counter()
with ThreadPoolExecutor(16) as pool:
    tot = sum(len(x) for x in pool.map(sam._bgzf_block, blocks))
print("host zlib-6, 16 threads: %.3f s, %d bytes" % (time.perf_counter() - t0, tot))
