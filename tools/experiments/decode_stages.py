import os, sys, time, tempfile, pathlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mapdamage_amd import engine, sam, synth
ref = synth.make_genome()
n = 8_000_000
batch = synth.parallel_batch("config3_batch", ref, n, seed=3003, workers=64)
tmp = tempfile.mkdtemp(prefix="mdx_dab_")
path = os.path.join(tmp, "c3.bam")
sam.write_bam(path, batch, ref.names, ref.lengths, [{"ID": "rg1", "SM": "synthetic", "LB": "lib1"}], rg_of_record=["rg1"] * batch.n, workers=64)
with open(path, "rb") as fh:
    while fh.read(1 << 26): pass
with engine.DamageEngine([("synthetic", "lib1")], 70, 10, 0) as eng:
    eng.set_reference(ref)
    for rep in range(4):
        sync = rep >= 2
        eng.reset(); eng.sync()
        t0 = time.perf_counter(); marks = []
        g = sam.GpuBamStream(eng, path, readgroups=[("rg1", 0)], chunk_bytes=256 << 20)
        g.__enter__()
        marks.append(("open", time.perf_counter()))
        while True:
            v = g.next_view()
            if sync: eng.sync()
            marks.append(("next", time.perf_counter()))
            if v is None: break
            eng.tabulate_view(v)
            if sync: eng.sync()
            marks.append(("tab", time.perf_counter()))
        got = eng.finish()
        marks.append(("finish", time.perf_counter()))
        g.__exit__(None, None, None)
        marks.append(("close", time.perf_counter()))
        prev = t0; out = []
        for k, t in marks:
            out.append("%s %.1f" % (k, (t - prev) * 1e3)); prev = t
        print("rep", rep, "sync" if sync else "async", "total %.1f ms:" % ((marks[-1][1] - t0) * 1e3), " ".join(out), flush=True)
