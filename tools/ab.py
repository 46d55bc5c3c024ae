"""A/B of library builds on the GPU box: kernel-only time of the split-cost workloads for each build
(tools/bin/libmdx_<tag>.so; "cur" = the in-tree build), batches generated once; optionally the fuzz against the oracle
per build.  usage: python tools/ab.py [--reads N] [--variants "a|b"] [--fuzz K] [--reps R] tag1 tag2 ..."""
import argparse
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import engine, synth  # noqa: E402
from tools.split_cost import VARIANTS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tags", nargs="+")
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--variants", default="plain paired|softclip 10%|ins 8%|del 8%|skip 0.2%|config 3|len 35-69|config 4")
    ap.add_argument("--fuzz", type=int, default=0)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--length", type=int, default=70)
    ap.add_argument("--packed", action="store_true", help="resident SEQ columns in their 4-bit form (the packed kernel)")
    args = ap.parse_args()
    only = [v for v in args.variants.split("|") if v]
    engine.DamageEngine.default_packed = args.packed
    ref = synth.make_genome()
    batches = {name: synth.parallel_batch(dict(dict(read_len=100, paired=True, contigs=[0, 1]), **kw), ref, args.reads, 3, workers=64)
               for name, kw in VARIANTS if name in only}
    out = ROOT / "gpurun_out" / "ab"
    out.mkdir(parents=True, exist_ok=True)
    for tag in args.tags:
        path = ROOT / "mapdamage_amd" / "libmdx.so" if tag == "cur" else ROOT / "tools" / "bin" / ("libmdx_%s.so" % tag)
        engine._lib = engine.load_library(str(path))
        rows = []
        with engine.DamageEngine([("s", "l")], args.length, 10, 0, lgd_max=4096) as eng:
            eng.set_reference(ref)
            for name in only:
                db = eng.upload(batches[name])
                eng.tabulate(db)
                eng.sync()
                eng.timing(True)
                for _ in range(args.reps):
                    eng.tabulate(db)
                eng.sync()
                n_launch, ms = eng.timing_read()
                eng.timing(False)
                db.free()
                rows.append({"variant": name, "reads": args.reads, "kernel_ms": ms / n_launch,
                             "ms_per_2M": ms / n_launch * 2e6 / args.reads})
        (out / (tag + ".jsonl")).write_text("".join(json.dumps(r) + "\n" for r in rows))
        print(tag, "ms/2M:", " | ".join("%s %.4f" % (r["variant"][:12], r["ms_per_2M"]) for r in rows), flush=True)
        if args.fuzz:
            from tools import fuzz_gpu
            sys.argv = ["fuzz_gpu", str(args.fuzz)]
            try:
                fuzz_gpu.main()
            except SystemExit as e:
                print(tag, "fuzz:", "ok" if not e.code else "DIFFERENCES", flush=True)


if __name__ == "__main__":
    main()
