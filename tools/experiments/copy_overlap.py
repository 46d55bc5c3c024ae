"""Would the fused launch gain from leaving the copy of the quality column to a copy on a second stream?  The fused kernel of
a build without its copy (-DMDX_RSABL_NOCOPY: wrong qualities, right time) alone, a device-to-device copy of the column
alone, and both at once (the copy of step k under the kernel of step k + 1).
    tools/mkvariant.sh nocopy -DMDX_RSABL_NOCOPY
    gpurun -- 'MDX_LIB=tools/bin/libmdx_nocopy.so python tools/experiments/copy_overlap.py 25000000'"""
import os, sys, pathlib, time
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch
torch.cuda.init()
from mapdamage_amd import engine, synth
import bench
if os.environ.get('MDX_LIB'):
    engine._lib = engine.load_library(os.environ['MDX_LIB'])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
ref = synth.make_genome()
model, corr = bench.rescale_fixture()
b = bench.add_mates(synth.parallel_batch(bench.CONFIG5, ref, n, 3, workers=64), 7)
dev = torch.device("cuda", 0)
with engine.DamageEngine([("s", "l")], 70, 10, 0, lgd_max=4096) as eng:
    eng.set_reference(ref)
    eng.set_rescale_model(model)
    db = eng.upload(b, packed=True)
    rs = bench.RescaleBuffers(torch, dev, b)
    src = torch.empty(b.seq.shape[0] + 64, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    side = torch.cuda.Stream(device=dev)
    K = 20
    def timed(kernel, copy):
        for _ in range(3):
            if kernel: rs.run(eng, db)
            if copy:
                with torch.cuda.stream(side): dst.copy_(src, non_blocking=True)
        eng.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            if kernel: rs.run(eng, db)
            if copy:
                with torch.cuda.stream(side): dst.copy_(src, non_blocking=True)
        eng.sync(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3
    print("records %d, quality column %.2f GB" % (n, src.numel() / 1e9))
    print("fused launch alone      %.3f ms per step" % timed(True, False))
    print("copy alone              %.3f ms per step" % timed(False, True))
    print("both, on two streams    %.3f ms per step" % timed(True, True))
    db.free()
